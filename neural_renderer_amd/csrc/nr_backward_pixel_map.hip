// nr_backward_pixel_map.hip -- K6, the approximate gradient of rgb / alpha w.r.t. vertex x, y
// (reference Rasterize.backward_pixel_map_gpu, rasterize.py:517-748) + its C-ABI entry point.
#include "nr_device.h"
#include "nr_band_lines.h"
#include "nr_k6_tune.h"

#include <atomic>
#include <mutex>
#include <type_traits>

using namespace nr;

namespace {

// --------------------------------------------------------------------------------------------------
// B1: backward_pixel_map (rasterize.py:517-748), global-memory form.  This kernel is the FALLBACK (raster sizes whose
// bands do not fit in LDS, or NR_FLAG_K6_GLOBAL); the default path is the band pipeline further down.
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
__global__ __launch_bounds__(WAVE) void k_bpm_global(
    const float *__restrict__ faces, const int32_t *__restrict__ fi_map, const float *__restrict__ rgb_map,
    const float *__restrict__ alpha_map, const float *__restrict__ g_rgb, const float *__restrict__ g_alpha,
    float *__restrict__ grad_faces, int F, int S, double eps)
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
    if (lane_on && p0x != p1x && d0_to >= d0_from) n_lines = d0_to - d0_from + 1;
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


// ==================================================================================================
// Band pipeline (default path).  Measurements on the headline scene (profiles/r01b) showed the global-memory
// kernel above to be bound by memory latency and, for the vertical sweeps (axis 0, stride S), by 3x
// cache-line amplification.  The band pipeline makes every sweep an LDS access:
//
//   [k_mark_visible]   only when the forward did not hand over its per-face "owns a pixel" flags (visible_faces)
//   k_compact_par | k_count_visible + k_compact_visible   per image, the sorted list of faces that own at least one
//          pixel.  A face that owns no pixel contributes nothing to K6 (the out sweep needs face_index[in] == fn,
//          :604, the in sweep only counts pixels owned by fn, :707), so ~2/3 of the front faces drop out.  The
//          compaction also stores, per listed face, the line range of each of its 3 edges along both axes, zeroes the
//          face's six double sums (indexed by list position: no fill launch), records face -> list position and counts the
//          lines of every (axis, band): a band workgroup whose count is zero leaves at once (more than half of the bands
//          of a teapot view: the object covers 12 % of the image).
//   k_line_setup   every line's record (crossing point, in / out pixels, sweep ranges, the two distance coefficients:
//          rasterize.py:573-579, :606-609, :665-672), written band by band into one buffer.
//          (The records and the kernel's body live in nr_band_lines.h: in the fused backward of a small call the K7 / K8
//          gather shares this launch -- nr_backward_gather.hip, k_setup_gather, through the SetupHook of
//          run_backward_pixel_map.)
//   k_bpm_fast<RGB, ALPHA, MODE>   one workgroup per (image, axis, band of W consecutive lines d0); workgroup ids are
//          mapped so that all bands of an image run on one XCD (xcd_block).  It
//          1. stages the band's W x S pixels in LDS, laid out [line][d1] so that a sweep is a contiguous LDS run
//             whatever the axis;
//          2. takes its line records a window at a time: the band's slice of k_line_setup's buffer -- or, for an image whose
//             records exceed the buffer (and with NR_FLAG_K6_SCAN), a scan of the image's visible faces (one per thread, 12
//             coalesced bytes each: the precomputed d0 ranges clipped to the band) that sets the window's lines up in place;
//          3. sweeps (fast_sweeps): the in / out sweeps of all lines are cut into pieces of <= FSEG = 15 pixels, sorted into
//             coverage classes, numbered through by one packed scan and handed to the threads through a descriptor queue;
//             the two partial sums of a piece are added to per-line LDS accumulators (ds_add_f64);
//          4. adds the line sums to a double scratch array [B][list position][3 vertices][x|y] (global_atomic_add_f64).
//   k_bpm_finalize   rounds the scratch sums to float and STORES grad_faces (z = 0; zeros for unlisted faces).
//
// One kernel, two arithmetic modes for the per-pixel terms (DESIGN.md "K6 numerics"; the sweep structure -- which pixels are
// visited, by which thread, in which class -- is the same):
//   K6_FAST   the north star's tolerance (1e-4) spent where it buys time: fused multiply-adds, v_rcp_f32, float piece sums.
//   K6_EXACT / K6_EXACT_POW2 (NR_FLAG_EXACT_GRADIENT)   every per-pixel term with the reference's arithmetic (its operations
//          one by one, IEEE division, the double `dist +- eps`), all sums in double: the correctly rounded sum of the
//          reference's terms up to double round-off (<= 2e-6 against the exactly summed oracle).  _POW2: S is a power of two
//          (x * 2. / S is then one exact float multiply; the generic form carries a double-precision division).
constexpr int BAND_WIN = 256;    // line records per window, at most
constexpr int FSEG = 15;         // pixels per piece (odd: consecutive pieces of a sweep start on different LDS banks;
                                 // re-swept in round 3: 9 / 12 / 15 / 18 pixels -> stage 242 / 237 / 230 / 233 us)
enum { K6_FAST = 0, K6_EXACT_POW2 = 1, K6_EXACT = 2 };

__global__ __launch_bounds__(256) void k_mark_visible(const int32_t *__restrict__ fi_map,
                                                      unsigned char *__restrict__ flags, int F, int SS, size_t P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int fi = fi_map[i];
    if (fi >= 0) flags[(i / SS) * F + fi] = 1;
}

// What the compaction records for the face at list position `pos` of image b: its index, the six edge line ranges
// rng[b][axis][pos][edge] (so that the 2 * n_bands band workgroups of an image scan 12 coalesced bytes per face instead of
// chasing list -> vertices each) and six zeroed double sums.
__device__ __forceinline__ void emit_visible(int b, int fn, int pos, int F, int S, const float *__restrict__ faces,
                                             int *__restrict__ vis_list, unsigned *__restrict__ rng,
                                             double *__restrict__ scratch, int *band_count, int n_bands, int W,
                                             int *line_diff = nullptr)
{
    vis_list[(size_t)b * F + pos] = fn;
    const float *f = faces + ((size_t)b * F + fn) * 9;
    const float fs = (float)S;
    float px[3], py[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { px[k] = to_pixel(f[3 * k], fs); py[k] = to_pixel(f[3 * k + 1], fs); }
    unsigned *r0 = rng + (((size_t)b * 2 + 0) * F + pos) * 3, *r1 = rng + (((size_t)b * 2 + 1) * F + pos) * 3;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const unsigned ra = edge_range(px[e], px[(e + 1) % 3], S), rb = edge_range(py[e], py[(e + 1) % 3], S);
        r0[e] = ra;
        r1[e] = rb;
        // lines per (axis, band of W lines): a band workgroup whose count is zero leaves at once
#pragma unroll
        for (int axis = 0; axis < 2; axis++) {
            const unsigned r = axis ? rb : ra;
            const int lo = (int)(r & 0xffffu), hi = (int)(r >> 16);
            if (line_diff) {
                // (k_compact_par: +1 where the edge's lines begin, -1 behind their end; the workgroup's prefix sum turns the
                // array into "edges on this line", whose sums over a band's lines are the counts -- two atomics per edge and
                // axis instead of one per band crossed, a loop of ~5 dependent LDS atomics on one-line bands)
                if (lo <= hi) {
                    atomicAdd(line_diff + axis * (S + 1) + lo, 1);
                    atomicAdd(line_diff + axis * (S + 1) + hi + 1, -1);
                }
                continue;
            }
            for (int band = lo / W; band * W <= hi; ++band)  // lo > hi (RNG_EMPTY): no iteration
                atomicAdd(band_count + axis * n_bands + band, min(hi, band * W + W - 1) - max(lo, band * W) + 1);
        }
    }
    double2 *z = reinterpret_cast<double2 *>(scratch + ((size_t)b * F + pos) * 6);
    z[0] = z[1] = z[2] = make_double2(0.0, 0.0);
}

// grad_faces of a face that owns no pixel (K6 contributes nothing, rasterize.py:604 / :707; K8 neither): when the fused
// backward finishes K6 inside its gather launch, the compaction -- which visits every face anyway -- stores these zeros
// (zero_listed: those of the listed faces too -- the fused backward whose gather runs BEFORE the band kernel and adds K8's sums)
__device__ __forceinline__ void zero_face(float *__restrict__ o)
{
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = 0.0f;
}

// Ordered compaction: every face gets slot_of = its position in the image's sorted list of visible faces, or -1.  One
// workgroup per chunk of 1024 faces.
//   Meshes of up to SMALL_CHUNKS chunks, one launch (k_compact_par): a workgroup counts the flags of the chunks in front of
//   its own (<= 15 coalesced bytes per thread) and leaves its lines-per-band counts as a dense row chunk_band[b][chunk][.];
//   the consumer (k_line_setup, or k_band_total when there is none) adds the rows up and takes the prefix.  (One workgroup
//   per image walking the chunks in order took 16 us for 5 chunks, 99 us for a 2048 x 2048 view with its 2 x 2048 bands.)
//   Larger meshes (config 5: 655 360 faces, B = 1), two launches + k_band_scan: k_count_visible counts the flags of each
//   chunk, k_compact_visible turns the counts of the preceding chunks into the chunk's offset and adds its band counts to
//   the image's global counters.
constexpr int VIS_CHUNK = 1024;
constexpr int SMALL_CHUNKS = 16;

__global__ __launch_bounds__(VIS_CHUNK) void k_compact_par(const unsigned char *__restrict__ flags,
                                                           int *__restrict__ vis_list, int *__restrict__ vis_count,
                                                           int *__restrict__ slot_of, int F, int n_chunks,
                                                           const float *__restrict__ faces, unsigned *__restrict__ rng,
                                                           double *__restrict__ scratch, int S,
                                                           int *__restrict__ chunk_band, int n_bands, int W,
                                                           int *__restrict__ band_cursor, float *__restrict__ zero_faces,
                                                           int zero_listed, int use_diff)
{
    extern __shared__ int s_band[];  // [2][n_bands] lines per band of this chunk's faces; use_diff: + [2][S + 1] line differences
    __shared__ int s_wcnt[VIS_CHUNK / 64];
    __shared__ int s_part[VIS_CHUNK / 64];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * n_bands; i += VIS_CHUNK) s_band[i] = 0;
    int *s_diff = use_diff ? s_band + 2 * n_bands : nullptr;
    if (use_diff)
        for (int i = tid; i < 2 * (S + 1); i += VIS_CHUNK) s_diff[i] = 0;
    int part = 0;  // (per wave) visible faces of this wave's columns of the preceding chunks
    for (int c = 0; c < chunk; ++c) part += __popcll(__ballot(flags[(size_t)b * F + c * VIS_CHUNK + tid] != 0));
    const int fn = chunk * VIS_CHUNK + tid;
    const bool v = fn < F && flags[(size_t)b * F + fn] != 0;
    const unsigned long long m = __ballot(v);
    if (lane == 0) { s_part[wave] = part; s_wcnt[wave] = __popcll(m); }
    __syncthreads();
    int off = 0, own = 0;
    for (int w = 0; w < VIS_CHUNK / 64; ++w) {
        off += s_part[w];
        if (w < wave) off += s_wcnt[w];
        own += s_wcnt[w];
    }
    if (fn < F) {
        const int before = off + __popcll(m & ((1ull << lane) - 1ull));  // visible faces in front of fn
        slot_of[(size_t)b * F + fn] = v ? before : -1;
        if (v) emit_visible(b, fn, before, F, S, faces, vis_list, rng, scratch, s_band, n_bands, W, s_diff);
        if (zero_faces && (!v || zero_listed)) zero_face(zero_faces + ((size_t)b * F + fn) * 9);
    }
    __syncthreads();
    if (use_diff) {
        // line differences -> edges per line (inclusive prefix over the 2 (S + 1) entries; the entry behind an axis' last line
        // takes the closing -1s, so the running sum is back at 0 where the second axis begins) -> lines per band
        const int n2 = 2 * (S + 1), per = (n2 + VIS_CHUNK - 1) / VIS_CHUNK, i0 = tid * per, i1 = min(n2, i0 + per);
        int local = 0;
        for (int i = i0; i < i1; ++i) local += s_diff[i];
        int inc = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, WAVE);
            if (lane >= o) inc += t;
        }
        __syncthreads();  // (s_part / s_wcnt were read above by every thread: reuse s_wcnt for the wave totals)
        if (lane == 63) s_wcnt[wave] = inc;
        __syncthreads();
        int run = inc - local;
        for (int w = 0; w < wave; ++w) run += s_wcnt[w];
        for (int i = i0; i < i1; ++i) {
            run += s_diff[i];
            s_diff[i] = run;
        }
        __syncthreads();
        for (int i = tid; i < 2 * n_bands; i += VIS_CHUNK) {
            const int axis = i >= n_bands, band = i - axis * n_bands;
            int c = 0;
            for (int l = band * W; l < min(S, band * W + W); ++l) c += s_diff[axis * (S + 1) + l];
            s_band[i] = c;
        }
        __syncthreads();
    }
    int *row = chunk_band + ((size_t)b * n_chunks + chunk) * 2 * n_bands;
    for (int i = tid; i < 2 * n_bands; i += VIS_CHUNK) row[i] = s_band[i];
    if (chunk == 0)  // the fill cursors of k_line_setup (it runs after this kernel)
        for (int i = tid; i < 2 * n_bands; i += VIS_CHUNK) band_cursor[(size_t)b * 2 * n_bands + i] = 0;
    if (chunk == n_chunks - 1 && tid == 0) {
        int base = 0;
        for (int w = 0; w < VIS_CHUNK / 64; ++w) base += s_part[w];
        vis_count[b] = base + own;
    }
}

// the lists alone (depth-only backward: no K6, but the K8 gather still wants to visit only the faces that own a pixel)
__global__ __launch_bounds__(VIS_CHUNK) void k_list_visible(const unsigned char *__restrict__ flags,
                                                            int *__restrict__ vis_list, int *__restrict__ vis_count, int F,
                                                            int n_chunks)
{
    __shared__ int s_wcnt[VIS_CHUNK / 64];
    __shared__ int s_part[VIS_CHUNK / 64];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int part = 0;
    for (int c = 0; c < chunk; ++c) part += __popcll(__ballot(flags[(size_t)b * F + c * VIS_CHUNK + tid] != 0));
    const int fn = chunk * VIS_CHUNK + tid;
    const bool v = fn < F && flags[(size_t)b * F + fn] != 0;
    const unsigned long long m = __ballot(v);
    if (lane == 0) { s_part[wave] = part; s_wcnt[wave] = __popcll(m); }
    __syncthreads();
    int off = 0, own = 0;
    for (int w = 0; w < VIS_CHUNK / 64; ++w) {
        off += s_part[w];
        if (w < wave) off += s_wcnt[w];
        own += s_wcnt[w];
    }
    if (v) vis_list[(size_t)b * F + off + __popcll(m & ((1ull << lane) - 1ull))] = fn;
    if (chunk == n_chunks - 1 && tid == 0) {
        int base = 0;
        for (int w = 0; w < VIS_CHUNK / 64; ++w) base += s_part[w];
        vis_count[b] = base + own;
    }
}

__global__ __launch_bounds__(VIS_CHUNK) void k_count_visible(const unsigned char *__restrict__ flags,
                                                             int *__restrict__ chunk_count, int F, int n_chunks,
                                                             int *__restrict__ band_lines, int n_bands)
{
    __shared__ int s_wcnt[VIS_CHUNK / 64];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the image's band counters are accumulated with global atomics by k_compact_visible: zero them here (one launch earlier)
    for (int i = chunk * VIS_CHUNK + tid; i < 2 * n_bands; i += n_chunks * VIS_CHUNK) band_lines[(size_t)b * 2 * n_bands + i] = 0;
    const int fn = chunk * VIS_CHUNK + tid;
    const bool v = fn < F && flags[(size_t)b * F + fn] != 0;
    const unsigned long long m = __ballot(v);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < VIS_CHUNK / 64; ++w) tot += s_wcnt[w];
        chunk_count[(size_t)b * n_chunks + chunk] = tot;
    }
}

__global__ __launch_bounds__(VIS_CHUNK) void k_compact_visible(const unsigned char *__restrict__ flags,
                                                               const int *__restrict__ chunk_count,
                                                               int *__restrict__ vis_list, int *__restrict__ vis_count,
                                                               int *__restrict__ slot_of, int F, int n_chunks,
                                                               const float *__restrict__ faces,
                                                               unsigned *__restrict__ rng, double *__restrict__ scratch,
                                                               int S, int *__restrict__ band_lines, int n_bands, int W,
                                                               int lds_counters, float *__restrict__ zero_faces,
                                                               int zero_listed)
{
    // This workgroup's lines per band are counted in LDS and only the non-zero counters go to the image's global ones:
    // one device-wide atomic per (face, edge, band) cost 315 us on config 5 (same-address atomics from all 8 XCDs).
    extern __shared__ int s_band[];  // [2 * n_bands] when lds_counters
    __shared__ int s_wcnt[VIS_CHUNK / 64];
    __shared__ int s_part[VIS_CHUNK / 64];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (lds_counters)
        for (int i = tid; i < 2 * n_bands; i += VIS_CHUNK) s_band[i] = 0;
    // offset of this chunk = sum of the counts of the chunks before it
    int part = 0;
    for (int c = tid; c < chunk; c += VIS_CHUNK) part += chunk_count[(size_t)b * n_chunks + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, WAVE);
    const int fn = chunk * VIS_CHUNK + tid;
    const bool v = fn < F && flags[(size_t)b * F + fn] != 0;
    const unsigned long long m = __ballot(v);
    if (lane == 0) { s_part[wave] = part; s_wcnt[wave] = __popcll(m); }
    __syncthreads();
    int off = 0, own = 0;
    for (int w = 0; w < VIS_CHUNK / 64; ++w) {
        off += s_part[w];
        if (w < wave) off += s_wcnt[w];
        own += s_wcnt[w];
    }
    if (fn < F) {
        const int before = off + __popcll(m & ((1ull << lane) - 1ull));  // visible faces in front of fn
        slot_of[(size_t)b * F + fn] = v ? before : -1;
        if (zero_faces && (!v || zero_listed)) zero_face(zero_faces + ((size_t)b * F + fn) * 9);
        if (v)
            emit_visible(b, fn, before, F, S, faces, vis_list, rng, scratch,
                         lds_counters ? s_band : band_lines + (size_t)b * 2 * n_bands, n_bands, W);
    }
    if (lds_counters) {
        __syncthreads();
        for (int i = tid; i < 2 * n_bands; i += VIS_CHUNK) {
            const int c = s_band[i];
            if (c) atomicAdd(band_lines + (size_t)b * 2 * n_bands + i, c);
        }
    }
    if (chunk == n_chunks - 1 && tid == 0) {
        int base = 0;
        for (int w = 0; w < VIS_CHUNK / 64; ++w) base += s_part[w];
        vis_count[b] = base + own;
    }
}

// exclusive scan of one int per thread over the workgroup of NT threads; returns the exclusive prefix, *total = sum
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int *s_tmp, int *total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, WAVE);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) {
        const int c = s_tmp[w];
        if (w < wave) woff += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

// rasterize.py:651 / :656 / :722 / :727: `dist += eps` when 0 < dist, `dist -= eps` otherwise (eps is a double literal there).
// Returns +-eps; only the high dword depends on the comparison (one v_cndmask instead of two selects on 64-bit values).
__device__ __forceinline__ double signed_eps(float dist, unsigned eps_hi, unsigned eps_lo)
{
    return __hiloint2double((int)((0.0f < dist) ? eps_hi : (eps_hi ^ 0x80000000u)), (int)eps_lo);
}

__global__ __launch_bounds__(256) void k_line_setup(LineSetupArgs a) { line_setup_body(a, (int)blockIdx.x, (int)blockIdx.y); }

// exclusive prefix of an image's 2 * n_bands line counts (first wave of the block), the image's total and the verdict
// whether its records fit the buffer; zeroes the fill cursors
__device__ __forceinline__ void band_prefix(const int *cnt, int n2, int *__restrict__ start_out, int *__restrict__ cursor_out,
                                            int *__restrict__ ok_out, size_t cap)
{
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x, per = (n2 + 63) / 64;
    int local = 0;
    for (int i = lane * per; i < min(n2, (lane + 1) * per); ++i) local += cnt[i];
    int inc = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, WAVE);
        if (lane >= o) inc += t;
    }
    int run = inc - local;
    for (int i = lane * per; i < min(n2, (lane + 1) * per); ++i) {
        start_out[i] = run;
        cursor_out[i] = 0;
        run += cnt[i];
    }
    const int total = __shfl(inc, 63, WAVE);
    if (lane == 0) *ok_out = ((size_t)total <= cap) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_band_scan(const int *__restrict__ band_lines, int *__restrict__ band_start,
                                                   int *__restrict__ band_cursor, int *__restrict__ lines_ok, int n_bands,
                                                   size_t cap, int force_scan)
{
    const size_t o = (size_t)blockIdx.x * 2 * n_bands;
    band_prefix(band_lines + o, 2 * n_bands, band_start + o, band_cursor + o, lines_ok + blockIdx.x, force_scan ? 0 : cap);
}

// the same table from the chunk rows of k_compact_par when no k_line_setup follows (NR_FLAG_K6_SCAN): every
// image is told to take the scan path
__global__ __launch_bounds__(256) void k_band_total(const int *__restrict__ chunk_band, int n_sum, int *__restrict__ band_lines,
                                                    int *__restrict__ band_start, int *__restrict__ lines_ok, int n_bands)
{
    const int b = blockIdx.x, n2 = 2 * n_bands;
    const int *rows = chunk_band + (size_t)b * n_sum * n2;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        int t = 0;
        for (int c = 0; c < n_sum; ++c) t += rows[(size_t)c * n2 + i];
        band_lines[(size_t)b * n2 + i] = t;
        band_start[(size_t)b * n2 + i] = 0;
    }
    if (threadIdx.x == 0) lines_ok[b] = 0;
}

// --------------------------------------------------------------------------------------------------
// pieces of k_bpm_fast
struct FastPx {  // LDS pixel data of a band, [line][d1]
    int *fi;     // face index
    float *g;    // gradients (g_alpha, g_r, g_g, g_b) -- or g_alpha alone
    float *c;    // colours   (alpha, r, g, b)         -- or alpha alone
    float *bg;   // colour of the band's uncovered pixels
    unsigned *cov;  // coverage bits [line][CW words]: bit d1 & 31 of word d1 >> 5 set <=> a face owns pixel (line, d1)
    int CW;         // words per line
    int *span;      // [line][2]: first and last covered pixel of the line (first > last: none)
};

template <bool RGB, bool ALPHA, int NT>
__device__ __forceinline__ void fast_stage(const FastPx &px, const int32_t *__restrict__ fi_map,
                                           const float *__restrict__ rgb_map, const float *__restrict__ alpha_map,
                                           const float *__restrict__ g_rgb, const float *__restrict__ g_alpha, size_t img,
                                           int axis, int band_lo, int nld, int S, int SP)
{
    const int tid = threadIdx.x;
    // pixel d1 of band line ld.  The coverage bits (px.cov, zeroed by the caller before the barrier in front of this
    // function) let the out sweeps skip the face index: only the covered eighth of the pixels issues the atomic.
    auto put = [&](int ld, int d1, int fi, float al, float ga, float r, float g, float bl, float gr, float gg, float gb) {
        const int l = ld * SP + d1;
        px.fi[l] = fi;
        if (RGB) {
            *reinterpret_cast<float4 *>(px.g + 4 * (size_t)l) = make_float4(ga, gr, gg, gb);
            *reinterpret_cast<float4 *>(px.c + 4 * (size_t)l) = make_float4(al, r, g, bl);
            if (px.bg && fi < 0) *reinterpret_cast<float4 *>(px.bg) = make_float4(al, r, g, bl);  // same value from every such pixel
        } else {
            px.g[l] = ga;
            px.c[l] = al;
            if (px.bg && fi < 0) px.bg[0] = al;
        }
        if (px.cov && fi >= 0) atomicOr(px.cov + ld * px.CW + (d1 >> 5), 1u << (d1 & 31));  // (callers without coverage bits pass none)
    };
    // four adjacent pixels of a map row with one 16-byte load per field: the 4 columns of a vertical band (one thread per
    // row; pixel j -> line j, d1 = y) or 4 consecutive pixels of a horizontal band's line (one thread per quad; pixel j ->
    // line ld, d1 = x + j)
    auto put_quad = [&](size_t g, int ld0, int d10, int ld_step, int d1_step) {
        const int4 vf = *reinterpret_cast<const int4 *>(fi_map + g);
        float4 va = make_float4(0, 0, 0, 0), vg = va;
        if (ALPHA) {
            va = *reinterpret_cast<const float4 *>(alpha_map + g);
            vg = *reinterpret_cast<const float4 *>(g_alpha + g);
        }
        float rr[12], qq[12];
#pragma unroll
        for (int k = 0; k < 12; k++) rr[k] = qq[k] = 0.0f;
        if (RGB) {
            const float4 *pr = reinterpret_cast<const float4 *>(rgb_map + 3 * g);
            const float4 *pg = reinterpret_cast<const float4 *>(g_rgb + 3 * g);
            const float4 r0 = pr[0], r1 = pr[1], r2 = pr[2], q0 = pg[0], q1 = pg[1], q2 = pg[2];
            rr[0] = r0.x; rr[1] = r0.y; rr[2] = r0.z; rr[3] = r0.w; rr[4] = r1.x; rr[5] = r1.y; rr[6] = r1.z; rr[7] = r1.w;
            rr[8] = r2.x; rr[9] = r2.y; rr[10] = r2.z; rr[11] = r2.w;
            qq[0] = q0.x; qq[1] = q0.y; qq[2] = q0.z; qq[3] = q0.w; qq[4] = q1.x; qq[5] = q1.y; qq[6] = q1.z; qq[7] = q1.w;
            qq[8] = q2.x; qq[9] = q2.y; qq[10] = q2.z; qq[11] = q2.w;
        }
        const int fis[4] = {vf.x, vf.y, vf.z, vf.w};
        const float als[4] = {va.x, va.y, va.z, va.w}, gas[4] = {vg.x, vg.y, vg.z, vg.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            put(ld0 + j * ld_step, d10 + j * d1_step, fis[j], als[j], gas[j], rr[3 * j], rr[3 * j + 1], rr[3 * j + 2], qq[3 * j],
                qq[3 * j + 1], qq[3 * j + 2]);
    };
    if (axis && (S & 3) == 0) {  // a band line is an image row: thread -> (line, quad of 4 consecutive pixels)
        const int quads = S >> 2;
        for (int i = tid; i < nld * quads; i += NT) {
            const int ld = i / quads, x = (i - ld * quads) << 2;
            put_quad(img + (size_t)(band_lo + ld) * S + x, ld, x, 0, 1);
        }
    } else if (axis) {  // thread -> (line, x), x fastest (coalesced 4- and 12-byte loads)
        for (int i = tid; i < nld * S; i += NT) {
            const int ld = i / S, x = i - ld * S;
            const size_t g = img + (size_t)(band_lo + ld) * S + x;
            float al = 0, ga = 0, r = 0, gn = 0, bl = 0, gr = 0, gg = 0, gb = 0;
            if (ALPHA) { al = alpha_map[g]; ga = g_alpha[g]; }
            if (RGB) {
                r = rgb_map[3 * g]; gn = rgb_map[3 * g + 1]; bl = rgb_map[3 * g + 2];
                gr = g_rgb[3 * g]; gg = g_rgb[3 * g + 1]; gb = g_rgb[3 * g + 2];
            }
            put(ld, x, fi_map[g], al, ga, r, gn, bl, gr, gg, gb);
        }
    } else if (nld == 4 && (S & 3) == 0) {  // 4 adjacent columns: one thread per row
        for (int y = tid; y < S; y += NT) put_quad(img + (size_t)y * S + band_lo, 0, y, 1, 0);
    } else {  // generic columns: thread -> (row d1, line ld) with ld fastest
        for (int i = tid; i < nld * S; i += NT) {
            const int d1 = i / nld, ld = i - d1 * nld;
            const size_t g = img + (size_t)d1 * S + band_lo + ld;
            float al = 0, ga = 0, r = 0, gn = 0, bl = 0, gr = 0, gg = 0, gb = 0;
            if (ALPHA) { al = alpha_map[g]; ga = g_alpha[g]; }
            if (RGB) {
                r = rgb_map[3 * g]; gn = rgb_map[3 * g + 1]; bl = rgb_map[3 * g + 2];
                gr = g_rgb[3 * g]; gg = g_rgb[3 * g + 1]; gb = g_rgb[3 * g + 2];
            }
            put(ld, d1, fi_map[g], al, ga, r, gn, bl, gr, gg, gb);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// Step 3 of k_bpm_fast: the sweeps of the n_win line records in s_line, one PIECE (<= FSEG pixels of one sweep) per thread
// and round.  Pieces are sorted into classes, each walked by its own loop:
//   U  exactly FSEG pixels of an OUT sweep, none of them covered by a face (7 of 8 pixels of an out sweep are background, and
//      they come in long runs behind the silhouette): `I - ref` is the constant (background - reference colour), the visit
//      reads the four gradients and nothing else; unrolled over its 15 pixels with compile-time LDS offsets;
//   M  exactly FSEG pixels of an OUT sweep between the first and the last covered pixel of the sweep (the coverage bits say
//      where): every pixel's colour is read as well -- an uncovered one holds the background colour (K5);
//   G  everything else: the pieces (<= FSEG pixels) of the in sweeps -- ownership test :707, per-pixel sign of `+- eps` -- and
//      the remainder of the out sweep, the general loop; short pieces (<= G_SHORT pixels) are numbered apart from long ones.
// Piece -> thread without a search: the thread that owns line `tid` of the window classifies the line's pieces (coverage
// bits of the band, px.cov), one packed scan numbers the pieces of each class through (U first, then M, then G, every class
// starting on a multiple of 64 so that a wave never mixes classes), the owners write a (line | piece << 8) descriptor per
// piece into a queue, and after one barrier every thread walks the ids tid, tid + 512, ... of the queue.  The queue takes
// what the line window leaves of the workgroup's LDS (fast_band_config: ~2500 descriptors, a window's worth); a window with
// more pieces goes through it in rounds.  (The binary search over the lines' prefix sums that this replaced in round 3 cost
// ~100 instructions and 9 dependent LDS round trips per piece.)
// The two sums of a piece go to acc[2 * line + k] (ds_add_f64), k = 0 / 1 for the edge's first / second vertex.
//
// Arithmetic of a visit (rasterize.py:630-657 out sweep, :697-728 in sweep) by MODE:
//   K6_FAST   diff = sum_c (I_c - ref_c) * g_c in the reference's order, accumulated with fused multiply-adds; dist =
//      fma(c * 2/S, t, +-eps) in float, the sign of `+- eps` taken once per piece in U / M (t = d1 - d1_cross keeps its sign
//      beyond the crossing point, hence so do c0 * t and c1 * t, :650 / :655); diff * v_rcp_f32(dist); float sums over the
//      <= 15 terms of a piece and over the <= 16 pieces of a run of lanes, double from there on.  No branch in U / M:
//      dm = diff <= 0 ? 0 : diff (:647: a NaN diff is not `<= 0` and goes through).  nr_k6_tune.h holds the knobs.
//   K6_EXACT*   the reference's operations one by one (products and sums rounded separately, `x * 2. / is` and `dist +- eps`
//      in double as its literals make them, IEEE division), every sum in double.
// When a contribution is not taken (:648 / :653: d0 equals the vertex) its coefficient is +-Inf / NaN; the lane then
// accumulates garbage that is discarded after the loop (no per-visit test of the has0 / has1 flags).
static_assert(FSEG % k6::FB == 0, "batches must tile a piece");

// One pixel's four floats out of LDS as ONE 16-byte read.  In the colour-only instantiation the alpha slot is never used and the
// compiler narrows the load to ds_read2_b32 + ds_read_b32 (two LDS instructions at 4-byte granularity per pixel: that instance
// ran 15 % slower than the one that also handles alpha); keeping the first component formally alive keeps the b128 form.
__device__ __forceinline__ float4 lds_px4(const float *p)
{
    float4 v = *reinterpret_cast<const float4 *>(p);
    asm volatile("" : "+v"(v.x));
    return v;
}
constexpr int G_SHORT = 4;  // class G pieces up to this many pixels are numbered apart from the longer ones

// DPP moves inside a row of 16 lanes: value of the lane D places below / one place above; a lane without such a neighbour
// in its row, or whose neighbour is not executing, gets `old`
template <int D>
__device__ __forceinline__ int dpp_row_shr(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x110 + D, 0xf, 0xf, false); }
template <int D>
__device__ __forceinline__ float dpp_row_shr_v(float v)  // (0 where there is no such neighbour)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xf, 0xf, false));
}
template <int D>
__device__ __forceinline__ double dpp_row_shr_v(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + D, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + D, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int dpp_row_shl1(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x101, 0xf, 0xf, false); }

// The pieces of one line sit on neighbouring lanes (consecutive ids), and all of them add to the same two sums: 64 lanes on
// ~10 addresses make the LDS serialise an atomic per lane.  So the sums of each run of equal keys are formed with DPP moves
// first (a segmented scan inside the 16-lane rows; a run that crosses a row is flushed in two parts); on return only the last
// lane of a run holds a non-zero pair.  In float for the tolerance mode (<= 16 piece sums, a tree of depth 4; in double the
// moves and selects come in pairs: stage 249 vs 239 us at raster 256 in round 3), in double for the exact one.  Which pieces
// share a run depends on the order of the line records, so two calls of the tolerance mode agree to ~1e-6 of the largest
// gradient, not to the bit.  (All 64 lanes execute this: the id loop keeps the wave together.)
template <typename T>
__device__ __forceinline__ void run_sums(int key, T &p0, T &p1)
{
    const int k1 = dpp_row_shr<1>(-1, key), k2 = dpp_row_shr<2>(-1, key), k4 = dpp_row_shr<4>(-1, key),
              k8 = dpp_row_shr<8>(-1, key);
    T q0 = dpp_row_shr_v<1>(p0), q1 = dpp_row_shr_v<1>(p1);
    if (k1 == key) { p0 += q0; p1 += q1; }
    q0 = dpp_row_shr_v<2>(p0); q1 = dpp_row_shr_v<2>(p1);
    if (k2 == key) { p0 += q0; p1 += q1; }
    q0 = dpp_row_shr_v<4>(p0); q1 = dpp_row_shr_v<4>(p1);
    if (k4 == key) { p0 += q0; p1 += q1; }
    q0 = dpp_row_shr_v<8>(p0); q1 = dpp_row_shr_v<8>(p1);
    if (k8 == key) { p0 += q0; p1 += q1; }
    if (dpp_row_shl1(-1, key) == key) p0 = p1 = T(0);  // not the last lane of its run
}

// inclusive prefix sum over the 64 lanes with DPP moves only (no LDS crossbar round trips): Hillis-Steele inside each row of 16
// lanes, then the row totals are passed on (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ int wave_incl_sum_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return v;
}

// exclusive scan of one 64-bit word of four 16-bit counters per thread over the workgroup (no field overflows: every total
// is below 2^16); returns the exclusive prefix, *total = sum.  The two halves are scanned as ints with DPP moves: the LDS is
// the busiest unit of this kernel and a shuffle-based scan would go through its crossbar twelve times.
template <int NT>
__device__ __forceinline__ unsigned long long block_excl_scan64(unsigned long long v, unsigned long long *s_tmp,
                                                                unsigned long long *total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    const int ilo = wave_incl_sum_dpp(lo), ihi = wave_incl_sum_dpp(hi);
    const unsigned long long inc = (unsigned long long)(unsigned)ilo | ((unsigned long long)(unsigned)ihi << 32);
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    unsigned long long woff = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) {
        const unsigned long long c = s_tmp[w];
        if (w < wave) woff += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

template <bool RGB, bool ALPHA, int MODE, int NT>
__device__ __forceinline__ void fast_sweeps(const FastPx &px, const BandLine *s_line, int n_win, int SP, double eps, int S,
                                            double *acc, void *s_queue, int qcap, bool wide, unsigned long long *s_tmp)
{
    constexpr bool EXACT = MODE != K6_FAST;
    constexpr int FB = k6::FB;
    const int tid = threadIdx.x;
    // ---- owner of line `tid`: its pieces by class.  The covered pixels of an out sweep lie next to the edge it starts
    // from (the rest of the object), the background behind them: the full pieces from the one with the first covered pixel
    // to the one with the last are class M (pmin .. pmin + nM - 1; an uncovered gap between two covered stretches -- the
    // teapot's handle -- rides along), the others class U.  One pass over the line's coverage words finds both ends.
    int nF = 0, nM = 0, nG = 0, nS = 0, pmin = 0, nIn = 0, rin = 0, rem = 0;
    // (Owners are the first n_win threads.  Spreading them over all eight waves -- 24 lanes each instead of three full waves --
    // was measured and is slower, 261 vs 240 us: every LDS instruction of a sparsely filled wave costs the LDS what a full one
    // does, and the LDS is the unit these loops wait for.)
    const int my_line = tid;
    const bool owner = tid < n_win;
    if (owner) {
        const int in_rng = s_line[my_line].in_rng, out_rng = s_line[my_line].out_rng, geo = s_line[my_line].geo;
        const int il = (in_rng >> 16) - (in_rng & 0xffff) + 1, out_from = out_rng & 0xffff;
        const int ol = (out_rng >> 16) - out_from + 1;
        nF = ol > 0 ? ol / FSEG : 0;
        // class G pieces: nIn pieces of the in sweep (the last one rin pixels long) and the out remainder (rem pixels); the
        // short ones (<= G_SHORT pixels: nine in-sweeps in ten) are numbered apart from the long ones, so that the lanes of a
        // wave walk pieces of similar length
        nIn = il > 0 ? (il + FSEG - 1) / FSEG : 0;
        rin = il > 0 ? il - (nIn - 1) * FSEG : 0;
        rem = ol > 0 ? ol % FSEG : 0;
        nG = nIn + (rem > 0);
        nS = (rin > 0 && rin <= G_SHORT) + (rem > 0 && rem <= G_SHORT);
        if (nF > 0) {
            const int ld = (geo >> 16) & 0xff;
            const unsigned *cw = px.cov + ld * px.CW;
            // the pieces' pixels [out_from, last], clipped to the covered span of the band line
            const int last = out_from + nF * FSEG - 1;
            const int ca = max(out_from, px.span[2 * ld]), cb = min(last, px.span[2 * ld + 1]);
            // (four words per LDS read: the lines' word arrays start on 16 bytes and are padded to a multiple of four words)
            int cmin = 0x7fffffff, cmax = -1;
            if (ca <= cb) {
                for (int w4 = (ca >> 5) & ~3; w4 <= (cb >> 5); w4 += 4) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(cw + w4);
                    const unsigned qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int w = w4 + k;
                        unsigned m = qq[k];
                        if (w < (ca >> 5) || w > (cb >> 5)) m = 0u;
                        if (w == (ca >> 5)) m &= 0xffffffffu << (ca & 31);
                        if (w == (cb >> 5)) m &= 0xffffffffu >> (31 - (cb & 31));
                        if (m) {
                            cmin = min(cmin, 32 * w + __ffs((int)m) - 1);
                            cmax = max(cmax, 32 * w + 31 - __clz((int)m));
                        }
                    }
                }
            }
            if (cmax >= 0) {
                pmin = (cmin - out_from) / FSEG;
                nM = (cmax - out_from) / FSEG - pmin + 1;
            }
        }
    }
    // class U pieces travel in SUPER-PIECES of up to k6::U_GROUP consecutive pieces of one line (round 5): one descriptor, one
    // record decode and one flush for up to 45 pixels -- the pieces behind the silhouette come in long runs, and decode + flush
    // cost a piece about half of what its fifteen visits do.  The two runs of a line's U pieces (in front of its M pieces and
    // behind them) are grouped separately; a super-piece's count sits above its first piece number in the descriptor.
    // (tolerance mode only: with the exact mode's 62-slot visits the lanes of a wave that hold one piece wait three times as long
    // for those that hold three -- 342 -> 370 us)
    constexpr int UG = EXACT ? 1 : k6::U_GROUP;
    const int nUa = nF > 0 ? pmin : 0, nUb = nF - nM - nUa;  // U pieces in front of / behind the line's M pieces
    const int gUa = (nUa + UG - 1) / UG, gUb = (nUb + UG - 1) / UG;
    const int nU = gUa + gUb, nL = nG - nS;
    // one scan for the four numberings (16 bits each: a window holds < 2^16 pieces of every kind by the choice of win_lines)
    unsigned long long totals = 0;
    const unsigned long long offs = block_excl_scan64<NT>((unsigned long long)nU | ((unsigned long long)nM << 16) |
                                                          ((unsigned long long)nS << 32) | ((unsigned long long)nL << 48),
                                                      s_tmp, &totals);
    const int TU = (int)(totals & 0xffff), TM = (int)((totals >> 16) & 0xffff), TS = (int)((totals >> 32) & 0xffff),
              TL = (int)(totals >> 48);
    // class starts (multiples of 64): U at 0, M, G-short, G-long
    const int MA = (TU + 63) & ~63, SA = MA + ((TM + 63) & ~63), LA = SA + ((TS + 63) & ~63), total_ids = LA + TL;
    const int idU = (int)(offs & 0xffff), idM = MA + (int)((offs >> 16) & 0xffff), idS = SA + (int)((offs >> 32) & 0xffff),
              idL = LA + (int)(offs >> 48);
    unsigned short *q16 = reinterpret_cast<unsigned short *>(s_queue);
    unsigned *q32 = reinterpret_cast<unsigned *>(s_queue);
    // arithmetic constants of the mode
    const float eps_f = (float)eps;                                                      // K6_FAST
    const unsigned eps_hi = (unsigned)__double2hiint(eps), eps_lo = (unsigned)__double2loint(eps);  // K6_EXACT*
    const double s_d = (double)S;
    const float two_over_s_f = (float)(2.0 / (double)S);  // exact when S is a power of two
    // one term of the exact mode: :649-651 / :654-656 (x * 2. / S: an exact scaling when S is a power of two)
    auto exact_term = [&](float diff, float c, float t) {
        const float ct = c * t;
        float dist = MODE == K6_EXACT_POW2 ? ct * two_over_s_f : (float)((double)ct * 2.0 / s_d);
        dist = (float)((double)dist + signed_eps(dist, eps_hi, eps_lo));  // + eps when 0 < dist, - eps otherwise
        return diff / dist;
    };
    // rounds of qcap ids (qcap: a multiple of NT; most windows fit in one round)
    for (int lo = 0; lo < total_ids; lo += qcap) {
        const int hi = min(lo + qcap, total_ids);
        if (lo > 0) __syncthreads();  // the previous round's readers are done
        if (owner) {                  // descriptors (line | piece << 8) of this owner's pieces with ids in [lo, hi)
            auto put = [&](int id, int seg) {
                if (wide) q32[id - lo] = (unsigned)my_line | ((unsigned)seg << 8);
                else q16[id - lo] = (unsigned short)(my_line | (seg << 8));
            };
            for (int j = min(max(lo - idU, 0), nU), j1 = min(max(hi - idU, 0), nU); j < j1; ++j) {
                const int first = j < gUa ? j * UG : pmin + nM + (j - gUa) * UG;
                const int cnt = j < gUa ? min(UG, nUa - j * UG) : min(UG, nUb - (j - gUa) * UG);
                put(idU + j, first | ((cnt - 1) << (wide ? 22 : 6)));
            }
            for (int j = min(max(lo - idM, 0), nM), j1 = min(max(hi - idM, 0), nM); j < j1; ++j) put(idM + j, pmin + j);
            for (int j = 0, cs = 0, cl = 0; j < nG; ++j) {  // G piece j: in piece j (j < nIn) or the out remainder
                const int len = j < nIn - 1 ? FSEG : (j == nIn - 1 ? rin : rem);
                const int id = len <= G_SHORT ? idS + cs++ : idL + cl++;
                if (id >= lo && id < hi) put(id, j);
            }
        }
        __syncthreads();
        for (int id = lo + tid; rfl(id) < hi; id += NT) {  // (wave-uniform trip count: all 64 lanes stay together)
            const int wid = rfl(id);     // ids of a wave are 64 consecutive numbers from a multiple of 64: one class per wave
            const int cls = wid < MA ? 0 : (wid < SA ? 1 : 2);
            const int cls_end = min(hi, cls == 0 ? TU : (cls == 1 ? MA + TM : (wid < LA ? SA + TS : total_ids)));
            if (wid >= cls_end) continue;  // a wave of padding ids
            // a lane on a padding id behind its class walks piece 0 of line 0 (valid LDS addresses) and throws the result away
            const bool valid = id < cls_end;
            const unsigned desc = valid ? (wide ? q32[id - lo] : (unsigned)q16[id - lo]) : 0u;
            const int line = (int)(desc & 0xffu);
            int seg = (int)(desc >> 8), u_cnt = 1;
            if (cls == 0) {  // a super-piece of class U: count - 1 above the first piece number
                u_cnt = (wide ? (seg >> 22) : (seg >> 6)) + 1;
                seg &= wide ? 0x3fffff : 63;
            }
            const BandLine *L = &s_line[line];
            const int4 h = *reinterpret_cast<const int4 *>(L);
            const float4 c = *reinterpret_cast<const float4 *>(&L->cross);
            const int flags = (h.z >> 24) & 0xff;
            const int ld = (h.z >> 16) & 0xff, base = ld * SP;
            const int d1_in = h.z & 0xffff;
            const int out_from = h.y & 0xffff, out_to = h.y >> 16;
            // class G: which sweep and which pixels
            bool mode_in = false;
            int s_from = out_from + seg * FSEG, s_to = s_from + FSEG - 1;
            if (cls == 2) {
                const int in_from = h.x & 0xffff, in_to = h.x >> 16;
                const int il = in_to - in_from + 1;
                const int n_in = il > 0 ? (il + FSEG - 1) / FSEG : 0;
                mode_in = seg < n_in;
                if (mode_in) { s_from = in_from + seg * FSEG; s_to = min(s_from + FSEG - 1, in_to); }
                else { s_from = out_from + ((out_to - out_from + 1) / FSEG) * FSEG; s_to = out_to; }
            }
            // reference colour: the OUT sweep compares with the in pixel, the IN sweep with the out pixel
            const int lref = base + (mode_in ? d1_in + ((flags & 8) ? 1 : -1) : d1_in);
            float ra = 0.0f, rr = 0.0f, rg = 0.0f, rb = 0.0f;
            float ba = 0.0f, br = 0.0f, bgn = 0.0f, bb = 0.0f;  // colour of an uncovered pixel
            if (RGB) {
                const float4 q = *reinterpret_cast<const float4 *>(px.c + 4 * (size_t)lref);
                ra = q.x; rr = q.y; rg = q.z; rb = q.w;
                const float4 w = *reinterpret_cast<const float4 *>(px.bg);
                ba = w.x; br = w.y; bgn = w.z; bb = w.w;
            } else {
                ra = px.c[lref];
                ba = px.bg[0];
            }
            // diff = sum_c (I_c - ref_c) * g_c with the reference's operations in its order (:631-638 / :709-716; its leading
            // `0 +` only turns a -0 into +0, which no later step can tell apart).  An uncovered pixel has the background colour
            // (K5), whose difference to the reference colour is a constant of the piece.
            const float dba = ba - ra, dbr = br - rr, dbg = bgn - rg, dbb = bb - rb;
            const float cross = c.x, c0k = c.y, c1k = c.z;
            const int fnr = __float_as_int(c.w);
            // (tolerance mode: the dot product as fused multiply-adds -- 19 -> 13 VALU instructions per uncovered pixel with
            // the fused `c * t + eps` below, 23 -> 17 per covered one; `diff <= 0` may then decide differently from :647 where
            // diff is within ~3 roundings of 0, about a term of the size of a rounding.  The exact mode keeps the products and
            // sums apart.)
            constexpr bool FUSE = !EXACT && k6::FUSED_DIFF;
            auto bg_diff = [&](const float4 &g4, float ga) {
                if (!RGB) return dba * ga;
                float d;
                if constexpr (FUSE) {
                    d = ALPHA ? __builtin_fmaf(dbr, g4.y, dba * g4.x) : dbr * g4.y;
                    d = __builtin_fmaf(dbg, g4.z, d);
                    d = __builtin_fmaf(dbb, g4.w, d);
                } else {
                    d = ALPHA ? dba * g4.x + dbr * g4.y : dbr * g4.y;
                    d += dbg * g4.z;
                    d += dbb * g4.w;
                }
                return d;
            };
            auto own_diff = [&](const float4 &c4, float ca, const float4 &g4, float ga) {  // c4 / ca: the pixel's colour
                if (!RGB) return (ca - ra) * ga;
                float d;
                if constexpr (FUSE) {
                    d = ALPHA ? __builtin_fmaf(c4.y - rr, g4.y, (c4.x - ra) * g4.x) : (c4.y - rr) * g4.y;
                    d = __builtin_fmaf(c4.z - rg, g4.z, d);
                    d = __builtin_fmaf(c4.w - rb, g4.w, d);
                } else {
                    d = ALPHA ? (c4.x - ra) * g4.x + (c4.y - rr) * g4.y : (c4.y - rr) * g4.y;
                    d += (c4.z - rg) * g4.z;
                    d += (c4.w - rb) * g4.w;
                }
                return d;
            };
            // the reciprocal of the tolerance mode (nr_k6_tune.h: one Newton step brings v_rcp_f32's 1 ulp to ~0.5)
            auto recip = [&](float y) {
                float r = __builtin_amdgcn_rcpf(y);
                if constexpr (k6::NEWTON) r = __builtin_fmaf(__builtin_fmaf(-y, r, 1.0f), r, r);
                return r;
            };
            // the piece's two sums: double in the exact mode (and with NR_K6_BATCH_DOUBLE), float otherwise
            constexpr bool DSUM = EXACT || k6::BATCH_DOUBLE;
            typename std::conditional<DSUM, double, float>::type f0 = 0, f1 = 0;
            if (cls < 2) {
                // ---- U / M: FSEG pixels of an out sweep, unrolled; one address register per array, compile-time offsets
                const int l0 = base + s_from;
                const float d1f0 = (float)s_from;
                const float t_first = d1f0 - cross;
                const float e0 = (0.0f < c0k * t_first) ? eps_f : -eps_f, e1 = (0.0f < c1k * t_first) ? eps_f : -eps_f;
                const float *gp = px.g + (RGB ? 4 : 1) * (size_t)l0, *cp = px.c + (RGB ? 4 : 1) * (size_t)l0;
                // One visit without a branch: the visits of a batch are independent instruction chains that the scheduler
                // interleaves.  Tolerance mode: y is never 0 (x and its eps have one sign), so the reciprocal is finite
                // wherever the contribution is taken (:648 / :653) and 0 * it adds nothing.
                // d1fb: t of the piece's first pixel (fused dist: pixel k has t_first + k, both of one sign: no
                // cancellation) or its d1.  It is re-declared opaque per batch below: that keeps the compiler from computing
                // all FSEG values of t ahead of the loop, which costs a register each and pushed the kernel into spilling.
                constexpr bool T_INCR = !EXACT && k6::FUSED_DIST;
                float d1fb = T_INCR ? t_first : d1f0;
                float b0 = 0.0f, b1 = 0.0f;  // (NR_K6_BATCH_DOUBLE: float sums of one batch)
                auto visit = [&](float diff, int k) {
                    if constexpr (EXACT) {
                        const float t = (d1fb + (float)k) - cross;
                        const float q0 = exact_term(diff, c0k, t), q1 = exact_term(diff, c1k, t);
                        const bool skip = diff <= 0.0f;                            // :647 (a NaN diff goes through)
                        f0 -= (double)(skip ? 0.0f : q0);                          // :651
                        f1 -= (double)(skip ? 0.0f : q1);                          // :656
                    } else {
                        const float dm = (diff <= 0.0f) ? 0.0f : diff;
                        float y0, y1;
                        if constexpr (T_INCR) {
                            const float t = d1fb + (float)k;
                            y0 = __builtin_fmaf(c0k, t, e0); y1 = __builtin_fmaf(c1k, t, e1);  // :649-650 / :654-655
                        } else {
                            const float t = (d1fb + (float)k) - cross;
                            y0 = c0k * t + e0; y1 = c1k * t + e1;
                        }
                        if constexpr (k6::BATCH_DOUBLE) {
                            b0 = __builtin_fmaf(-dm, recip(y0), b0);
                            b1 = __builtin_fmaf(-dm, recip(y1), b1);
                        } else {
                            f0 = __builtin_fmaf(-dm, recip(y0), f0);                   // :651
                            f1 = __builtin_fmaf(-dm, recip(y1), f1);                   // :656
                        }
                    }
                };
                auto end_batch = [&]() {
                    if constexpr (!EXACT && k6::BATCH_DOUBLE) { f0 += (double)b0; f1 += (double)b1; b0 = b1 = 0.0f; }
                };
                if (cls == 0) {
                    // U: gradients only; the next batch's LDS reads are in flight while this one is evaluated.  The pieces of a
                    // super-piece one after the other: each piece's sums in a float of their own, added up like the run sums do
                    for (int u = 0; u < u_cnt; ++u) {
                    const auto p0 = f0, p1 = f1;
                    f0 = 0;
                    f1 = 0;
                    float4 gc[FB], gn[FB];
                    float ac[FB], an[FB];
#pragma unroll
                    for (int j = 0; j < FB; ++j) {
                        gc[j] = gn[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        ac[j] = an[j] = 0.0f;
                        if (RGB) gc[j] = lds_px4(gp + 4 * j);
                        else ac[j] = gp[j];
                    }
#pragma unroll
                    for (int kb = 0; kb < FSEG; kb += FB) {
                        asm volatile("" : "+v"(d1fb));
                        if (kb + FB < FSEG) {
#pragma unroll
                            for (int j = 0; j < FB; ++j) {
                                if (RGB) gn[j] = lds_px4(gp + 4 * (kb + FB + j));
                                else an[j] = gp[kb + FB + j];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < FB; ++j) visit(bg_diff(gc[j], ac[j]), kb + j);
                        end_batch();
#pragma unroll
                        for (int j = 0; j < FB; ++j) { gc[j] = gn[j]; ac[j] = an[j]; }
                        __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from hoisting every batch's reads to the top)
                    }
                    f0 += p0;
                    f1 += p1;
                    gp += (RGB ? 4 : 1) * FSEG;
                    d1fb += (float)FSEG;
                    }
                } else {
                    // M: every pixel's colour as well -- an uncovered one holds the background colour (K5), so the same
                    // expression serves both
#pragma unroll
                    for (int kb = 0; kb < FSEG; kb += FB) {
                        asm volatile("" : "+v"(d1fb));
                        float4 g4[FB], c4[FB];
                        float ga[FB], ca[FB];
#pragma unroll
                        for (int j = 0; j < FB; ++j) {
                            g4[j] = c4[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            ga[j] = ca[j] = 0.0f;
                            if (RGB) {
                                g4[j] = lds_px4(gp + 4 * (kb + j));
                                c4[j] = lds_px4(cp + 4 * (kb + j));
                            } else {
                                ga[j] = gp[kb + j];
                                ca[j] = cp[kb + j];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < FB; ++j) visit(own_diff(c4[j], ca[j], g4[j], ga[j]), kb + j);
                        end_batch();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                // ---- G: a piece of an in sweep or the remainder of an out sweep, the general loop
                const int own_mask = mode_in ? -1 : 0;
                float d1f = (float)s_from;
                float b0 = 0.0f, b1 = 0.0f;
                for (int l = base + s_from; l <= base + s_to; ++l, d1f += 1.0f) {
                    // face index, gradients and colour are requested together (one LDS round trip)
                    const int fi = px.fi[l];
                    float4 g4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c4 = g4;
                    float ga = 0.0f, ca = 0.0f;
                    if (RGB) { g4 = lds_px4(px.g + 4 * (size_t)l); c4 = lds_px4(px.c + 4 * (size_t)l); }
                    else { ga = px.g[l]; ca = px.c[l]; }
                    const float diff = own_diff(c4, ca, g4, ga);  // (an uncovered pixel holds the background colour)
                    // :707 (only the in sweep tests ownership) and :647 / :717 (a NaN diff is not `<= 0`), without divergent
                    // control flow on the sweep kind
                    if ((((fi ^ fnr) & own_mask) != 0) | (diff <= 0.0f)) continue;
                    const float t = d1f - cross;
                    if constexpr (EXACT) {
                        f0 -= (double)exact_term(diff, c0k, t);                               // :649-651
                        f1 -= (double)exact_term(diff, c1k, t);                               // :654-656
                    } else {
                        const float x0 = c0k * t, x1 = c1k * t;                               // :649 / :654 (2 / S folded into c)
                        const float y0 = x0 + ((0.0f < x0) ? eps_f : -eps_f);                 // :650 / :655
                        const float y1 = x1 + ((0.0f < x1) ? eps_f : -eps_f);
                        b0 = __builtin_fmaf(-diff, recip(y0), b0);                            // :651
                        b1 = __builtin_fmaf(-diff, recip(y1), b1);                            // :656
                    }
                }
                if constexpr (!EXACT) { f0 += b0; f1 += b1; }
            }
            // :648 / :653: a contribution whose vertex sits on the line is not taken (its coefficient was Inf / NaN)
            using RunT = typename std::conditional<EXACT || k6::RUNSUM_DOUBLE, double, float>::type;
            RunT p0 = (valid && (flags & 2)) ? (RunT)f0 : RunT(0), p1 = (valid && (flags & 4)) ? (RunT)f1 : RunT(0);
            const int key = valid ? line : -1 - (tid & 63);  // (a padding lane: a run of its own)
            run_sums(key, p0, p1);
            const double a0 = (double)p0, a1 = (double)p1;
            if (a0 != 0.0) atomicAdd(&acc[2 * line], a0);
            if (a1 != 0.0) atomicAdd(&acc[2 * line + 1], a1);
        }
    }
}

// The band kernel, NT threads per workgroup (band_shape below).  (launch bounds: what the LDS of a shape admits -- six waves
// per SIMD for three 512-thread workgroups per CU, four for four 256-thread ones; the generic exact form carries a double
// division: four)
template <bool RGB, bool ALPHA, int MODE, int NT, bool OVF = false>
__global__ __launch_bounds__(NT, MODE == K6_EXACT ? 4 : (NT == 256 ? k6::MINWAVES_256 : 6)) void k_bpm_fast(
    const float *__restrict__ faces, const int32_t *__restrict__ fi_map, const float *__restrict__ rgb_map,
    const float *__restrict__ alpha_map, const float *__restrict__ g_rgb, const float *__restrict__ g_alpha,
    const int *__restrict__ vis_list, const int *__restrict__ vis_count, const unsigned *__restrict__ rng,
    double *__restrict__ scratch, const int *__restrict__ band_lines, const int *__restrict__ band_start,
    const int *__restrict__ lines_ok, const BandLine *__restrict__ line_buf, size_t cap, int F, int S, int W, int SP,
    double eps, float k2s, int B, int win_lines, int qcap, uint4 *__restrict__ zero16, size_t n_zero16)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    // XCD-aware placement (nr_device.h): bands of one image share cache lines -- 8 adjacent 4-column bands sit in the same
    // 128-byte line of every map row, and the horizontal pass re-reads what the vertical pass just fetched -- so all
    // 2 * n_bands workgroups of an image are given ids that land on ONE XCD (one L2).  Measured in round 1: 613 -> 477 us.
    // (The same mapping on the forward and gather kernels changed nothing or cost 5 %: their reads are not shared.)
    const unsigned n_bands = (unsigned)(S + W - 1) / (unsigned)W;
    const unsigned total_wg = n_bands * 2u * (unsigned)B;
    // OVF: the launch behind k_bpm_row that serves the images whose records exceed the line buffer -- normally none -- by the scan
    // path: a small grid whose workgroups first take one cooperative look at the images' verdicts (none over the buffer: leave)
    // and otherwise walk the bands in strides.  (An instantiation of its own: with the band loop around it the ordinary kernel
    // came out at 116 registers instead of 89 and 7 % slower.)
    constexpr bool overflow_only = OVF;
    auto band_body = [&](const unsigned logical) {
    if (n_zero16) {
        // The fused backward's zero fill of grad_textures (the K7 gather behind this kernel stores only the listed faces'
        // cubes) rides along: every workgroup clears one slice before it looks at its band -- half of them have nothing
        // else to do -- instead of a launch of its own between this kernel and the gather.
        const size_t per = (n_zero16 + total_wg - 1) / total_wg, z_lo = (size_t)logical * per, z_hi = min(n_zero16, z_lo + per);
        for (size_t k = z_lo + tid; k < z_hi; k += NT) zero16[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    const int band = (int)(logical % n_bands), axis = (int)((logical / n_bands) & 1u), b = (int)(logical / (2u * n_bands));
    const int band_lo = band * W, band_hi = min(band_lo + W, S) - 1;
    const int nld = band_hi - band_lo + 1;
    const size_t bidx = ((size_t)b * 2 + axis) * n_bands + band;
    // overflow_only: this launch stands behind k_bpm_row and serves only the images whose records exceed the line buffer, by the
    // scan path; the band tables are k_bpm_row's (per line), not this kernel's shape
    if (overflow_only && lines_ok[b] != 0) return;
    const int n_band_lines = overflow_only ? 1 : band_lines[bidx];
    if (n_band_lines == 0) return;  // step 0: no visible face has a line here

    size_t off = 0;
    auto carve = [&](size_t bytes) { unsigned char *p = smem + off; off += (bytes + 15) & ~(size_t)15; return p; };
    constexpr int NC = RGB ? 4 : 1;  // floats per pixel in the gradient / colour arrays: (alpha, r, g, b) or alpha alone
    FastPx px;
    px.fi = (int *)carve((size_t)W * SP * 4);
    px.g = (float *)carve((size_t)W * SP * NC * 4);
    px.c = (float *)carve((size_t)W * SP * NC * 4);
    px.bg = (float *)carve(16);
    px.CW = (((SP + 31) >> 5) + 3) & ~3;  // words per line, a multiple of four (16-byte reads of the classification)
    px.cov = (unsigned *)carve((size_t)W * px.CW * 4);
    px.span = (int *)carve(4 * 2 * 4);  // (W <= 4 lines)
    const bool wide = S > 63 * FSEG;  // piece numbers beyond 6 bits (two more carry a class-U super-piece's count): 32-bit descriptors
    // The rest of the workgroup's LDS is split between the line window (32 B record + two double sums per line) and the piece
    // queue by the host (fast_band_config).  (A per-band split inside the kernel -- equal windows, as few as let a window's
    // pieces through the queue in one round -- was measured and lost to the fixed split, 258 vs 242 us: it trades windows of
    // 224 lines with two rounds for twice as many windows.)
    BandLine *s_line = (BandLine *)carve(sizeof(BandLine) * win_lines);
    double *s_lacc = (double *)carve(16 * (size_t)win_lines);  // [win_lines][2] sums of each line for its edge's two vertices
    void *s_queue = carve((size_t)qcap * (wide ? 4 : 2));     // piece descriptors of a window (or of a round of it)
    int *s_tmp = (int *)carve(4 * 16);

    // ---- 1. stage the band
    const size_t img = (size_t)b * S * S;
    for (int i = tid; i < W * px.CW; i += NT) px.cov[i] = 0u;
    __syncthreads();
    fast_stage<RGB, ALPHA, NT>(px, fi_map, rgb_map, alpha_map, g_rgb, g_alpha, img, axis, band_lo, nld, S, SP);
    __syncthreads();
    if (tid < nld) {  // covered span of each band line (the sweeps' classification looks no further)
        int first = 0x7fffffff, last = -1;
        for (int w = 0; w < px.CW; ++w) {
            const unsigned m = px.cov[tid * px.CW + w];
            if (m) { first = min(first, 32 * w + __ffs((int)m) - 1); last = 32 * w + 31 - __clz((int)m); }
        }
        px.span[2 * tid] = first;
        px.span[2 * tid + 1] = last;
    }

    // ---- 2. - 4. window by window.  Records path: the band's line records were written by k_line_setup.  Scan path (an image
    // with more lines than the record buffer holds, or NR_FLAG_K6_SCAN): the image's visible faces are scanned in chunks of
    // one face per thread and the lines of the chunk that fall into this band are set up in place.  The scan path keeps
    // nothing in registers across the sweeps -- it recomputes its chunk's line counts for every window -- so that the kernel's
    // register budget is the records path's.
    const bool use_rec = !overflow_only && lines_ok[b] != 0;
    const BandLine *recs = line_buf + (size_t)b * cap + (use_rec ? band_start[bidx] : 0);
    const int n_vis = use_rec ? 0 : vis_count[b];
    const unsigned *rng_ba = rng + ((size_t)b * 2 + axis) * F * 3;
    int chunk = 0, win = 0;  // (scan path) first list position of the chunk; (both) first line of the next window
    // the 2 * n_win line sums, one or two per thread (a window has at most NT lines)
    auto each_sum = [&](int n2, auto fn) {
        if (tid < n2) fn(tid);
        if (NT < 2 * BAND_WIN && tid + NT < n2) fn(tid + NT);
    };
    for (;;) {
        int n_win;
        if (use_rec) {
            if (win >= n_band_lines) break;
            n_win = min(n_band_lines - win, win_lines);
            if (tid < n_win) s_line[tid] = recs[win + tid];  // (win_lines <= NT)
            each_sum(2 * n_win, [&](int i) { s_lacc[i] = 0.0; });
            win += win_lines;
        } else {
            if (chunk >= n_vis) break;
            // one visible face per thread: lines of its 3 edges inside the band
            int nl = 0;
            int e_lo[3] = {0, 0, 0}, e_n[3] = {0, 0, 0};
            if (chunk + tid < n_vis) {
                const unsigned *r = rng_ba + (size_t)(chunk + tid) * 3;
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const unsigned pr = r[e];
                    const int lo = max((int)(pr & 0xffffu), band_lo), hi = min((int)(pr >> 16), band_hi);
                    if (hi >= lo) { e_lo[e] = lo; e_n[e] = hi - lo + 1; nl += e_n[e]; }
                }
            }
            int total_lines = 0;
            const int line_off = block_excl_scan<NT>(nl, s_tmp, &total_lines);
            if (win >= total_lines) {  // (uniform) this chunk is done
                chunk += NT;
                win = 0;
                continue;
            }
            n_win = min(total_lines - win, win_lines);
            // (list position | edge << 28, line) of the window's lines, parked where the line sums will be
            int2 *s_rec = reinterpret_cast<int2 *>(s_lacc);
            if (nl > 0 && line_off < win + win_lines && line_off + nl > win) {
                int k = line_off;
#pragma unroll
                for (int e = 0; e < 3; e++)
                    for (int j = 0; j < e_n[e]; j++, k++)
                        if (k >= win && k < win + win_lines) s_rec[k - win] = make_int2((chunk + tid) | (e << 28), e_lo[e] + j - band_lo);
            }
            __syncthreads();
            int2 rec = make_int2(0, 0);
            if (tid < n_win) rec = s_rec[tid];
            __syncthreads();
            if (tid < n_win) {  // line setup, one line per thread
                const int pos = rec.x & 0x0fffffff, e = (rec.x >> 28) & 3, ld = rec.y;
                const int rfn = vis_list[(size_t)b * F + pos];
                s_line[tid] = make_fast_line(faces + ((size_t)b * F + rfn) * 9, e, axis, band_lo + ld, ld, S, rfn,
                                             pos | (e << 28) | (((e + 1) % 3) << 30),
                                             [&](int d1) { return px.fi[ld * SP + d1]; }, k2s);
            }
            each_sum(2 * n_win, [&](int i) { s_lacc[i] = 0.0; });
            win += win_lines;
        }
        __syncthreads();
        fast_sweeps<RGB, ALPHA, MODE, NT>(px, s_line, n_win, SP, eps, S, s_lacc, s_queue, qcap, wide,
                                      reinterpret_cast<unsigned long long *>(s_tmp));
        __syncthreads();
        each_sum(2 * n_win, [&](int i) {  // line sums -> global double scratch [list position][vertex][x|y]
            const double a = s_lacc[i];
            if (a != 0.0) {
                const int tgt = s_line[i >> 1].tgt;
                const int pos = tgt & 0x0fffffff, v = (i & 1) ? (tgt >> 30) & 3 : (tgt >> 28) & 3;
                atomicAdd(scratch + ((size_t)b * F + pos) * 6 + 2 * v + (1 - axis), a);
            }
        });
        __syncthreads();
    }
    };
    if constexpr (!OVF) {
        const unsigned logical = xcd_block(total_wg);
        if (logical < total_wg) band_body(logical);
    } else {
        int any = 0;
        for (int i = tid; i < B; i += NT) any |= lines_ok[i] == 0;
        if (!__syncthreads_or(any)) return;
        for (unsigned logical = blockIdx.x; logical < total_wg; logical += gridDim.x) {
            band_body(logical);
            __syncthreads();  // (the next band re-uses the LDS)
        }
    }
}

// k_bpm_row: how many of a line's n_parts waves (a power of two) share one of its windows -- n_parts over the number of windows
// rounded up to a power of two, at least 1; lines of fewer than 32 segments (rasters below 481) keep a wave per window: their
// sweeps are too short for the members' repeated set-up to pay (64 views at 384^2: +1.7 % with teams, 512^2: -1.7 %)
__device__ __forceinline__ int window_team(int n_rec, int n_parts, int nsl)
{
    const int n_win = (n_rec + 63) >> 6;
    if (nsl < 32) return 1;
    int team = n_parts;
    while (team > 1 && n_parts / team < n_win) team >>= 1;
    return team;
}

// LDS hand-over between the lanes of ONE wave (its LDS operations execute in order; the fences keep the compiler from moving
// accesses across)
__device__ __forceinline__ void wave_lds_handover()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ==================================================================================================
// k_bpm_row (round 6): the band kernel of the default arithmetic mode.  A WAVE owns a band line, a BLOCK of 16 lanes a line
// record: four records per wave instruction.
//
// k_bpm_fast gives a lane a PIECE (15 pixels of one sweep) and walks it pixel by pixel out of LDS: every piece a descriptor, a
// record decode and a flush, every window a chain of barriers.  Round 5's k_bpm_px (gone) held a line's pixels in registers and
// visited ONE record at a time with all 64 lanes: 45 instructions of frame per record around its visits, and a sweep's last
// 64-lane chunk half empty.  Both ran at one third of the issued lanes useful.  Here:
//   * the band's pixels stay in LDS -- per pixel the four gradients g and ONE sum P = sum_c (I_c - K_c) g_c (formed in double at
//     the staging, rounded once), 24 bytes with the face index: five workgroups per CU.  K is the colour of the line's first
//     pixel, i.e. the background almost always: the sum of a background pixel is then exactly 0, and any colour NEAR the line's
//     colours serves -- with K = 0 a scene whose background is as bright as its faces (every term the small difference of
//     large products) came out 3 ... 6e-4 off in round 5;
//   * a block reads 16 consecutive pixels of its line per step, in segments ALIGNED to 16 pixels;
//   * an out sweep runs from the crossing point to the image border (:607-609), so "inside the sweep" is the sign of
//     td = direction * (d1 - d1_cross), a value the visit needs anyway: no range arithmetic, and a block may start segments
//     before its sweep or run on behind it -- those lanes are masked by the same comparison;
//   * the records of a window (64: a lane each in phase A) are SORTED by their number of segments (a counting sort in LDS) and
//     dealt to the blocks four at a time: the blocks of a group walk (nearly) the same number of steps (lane efficiency 0.89 at
//     raster 256, 0.92 at 512: scripts/row_stats.py), so the group is one loop with a uniform trip count and ONE reduction for
//     four records;
//   * that reduction is the matrix pipe's: a block is the 16 lanes of one block of v_mfma_f64_4x4x4_4b_f64 -- lanes
//     4 b .. 4 b + 3 of each of the wave's four rows -- and two of these instructions with a matrix of ones add up a block's 16
//     values in DOUBLE (the first contracts over the rows, the second over the four lanes), the sum arriving in every lane of
//     the block: no DPP tree in float (whose roundings of the TOTALS cost dense meshes a factor two in the error level),
//     no LDS atomics, and the vector pipe issues 4 conversions instead of 16 operations;
//   * a block's pixel quads are dealt to its four rows so that the 16-byte LDS reads stay conflict-free whatever segments the
//     four blocks are at (the LDS serves lanes {0-3, 12-15, 20-27}, ... together: quad = (row + 2 (b >> 1)) mod 4);
//   * a record's constants reach its block by ds_bpermute_b32 from the lane that prepared them in phase A.
// A visit (:630-657): diff = P - sum_c (ref_c - K_c) g_c (four fused multiply-adds), dist = |c| |t| + eps for the two vertices
// (t = d1 - d1_cross keeps its sign along an out sweep, so sigma = sign(c) sign(t) is a property of the record: the magnitudes
// are summed, the sign goes on once, at the flush), two v_rcp_f32 (a quarter-rate instruction on this part: a third of the
// visit's issue time, scripts/dev/valu_rate_probe.hip), two fused multiply-adds: 15 vector operations and two LDS reads for up
// to 64 useful lanes.
// Phase A (a lane per record): the record, its two reference colours straight from the maps (the in and the out pixel,
// :594-601 / :697-700 -- no colours in LDS), its in sweep (:665-728; nine in ten are <= 4 pixels) in the same two-sum form --
// and the two pixels next to the crossing point, which carry a record's largest terms, in the reference's DIRECT form
// sum_c (I_c - ref_c) g_c: their colours are the two reference colours (error levels, direct / centred sums for them: config 4
// 4.6e-5 / 5.9e-5, headline 3.3e-5 / 4.1e-5).
// Flush: one lane per record adds in sweep + out sweep to the double scratch (global_atomic_add_f64), as k_bpm_fast does.
// Images whose records exceed the line buffer (lines_ok == 0) are left to k_bpm_fast's scan path (launched behind this kernel
// with overflow_only set).
#ifndef NR_ROW_OFF  // (development, switch-off builds -- results wrong by construction: 1 no out sweeps, 2 no in sweeps, 4 no global atomics at the flush)
#define NR_ROW_OFF 0
#endif
#ifndef NR_ROW_LDS_PAD  // (development: unused LDS per workgroup, to probe what a workgroup less per CU costs)
#define NR_ROW_LDS_PAD 0
#endif
namespace rowk {
#ifndef NR_ROW_NT  // (development: threads per workgroup of k_bpm_row -- 256, 128 or 64)
#define NR_ROW_NT 256
#endif
constexpr int NT = NR_ROW_NT, NW = NT / 64;
constexpr int WIN = 64;            // records per window (a lane each in phase A)
constexpr int SEG = 16;            // pixels per step of a block
constexpr int MAX_SEGS = 64;       // segments of a line, at most (raster <= 1024): the sort's keys
constexpr int IN_SEG = 16;        // float terms per double addition of an in sweep (a piece of k_bpm_fast holds 15)
constexpr int IN_BATCH = 4;       // pixels of an in sweep whose LDS reads are requested together
#ifndef NR_ROW_MAX_PX  // (development)
#define NR_ROW_MAX_PX 1024
#endif
constexpr int MAX_PX = NR_ROW_MAX_PX;       // pixels of a band, at most
// LDS of a workgroup, for the npx = W * SP pixels of its band: gradients [W][SP][NC] | sums P [W][SP] (the exact mode: colours
// [W][SP][NC]) | face indices [W][SP] | K of the band's lines (not in the exact mode) | a window per wave
constexpr int K_BYTES = 4 * 16, WAVE_BYTES = WIN * 16;
template <bool RGB> __host__ __device__ constexpr size_t g_bytes(size_t npx) { return npx * (RGB ? 16 : 4); }
template <bool RGB, bool EXACT> __host__ __device__ constexpr size_t c_bytes(size_t npx) { return EXACT ? g_bytes<RGB>(npx) : npx * 4; }
template <bool RGB, bool EXACT> __host__ __device__ constexpr size_t lds_bytes(size_t npx)
{
    return g_bytes<RGB>(npx) + c_bytes<RGB, EXACT>(npx) + npx * 4 + (EXACT ? 0 : K_BYTES) + NW * WAVE_BYTES + NR_ROW_LDS_PAD;
}
}  // namespace rowk

#ifdef NR_ROW_STATS  // development builds (scripts/row_stats.py): work counters of k_bpm_row
__device__ unsigned long long g_row_stats[8];
NR_API int nr_dev_row_stats(unsigned long long *out8)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_row_stats), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_row_stats), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#define NR_ROW_STAT(i, v) do { if (lane == 0) atomicAdd(&g_row_stats[i], (unsigned long long)(v)); } while (0)
// (the records of the first NR_ROW_DUMP_WINDOWS windows -- segments of the out sweep | direction << 8 | has an out sweep << 9 |
// valid << 10 -- for studies of the grouping on the host: scripts/row_groupings.py)
constexpr unsigned NR_ROW_DUMP_WINDOWS = 1u << 16;
__device__ unsigned g_row_dump_n;
__device__ unsigned g_row_dump[NR_ROW_DUMP_WINDOWS * 64];
NR_API int nr_dev_row_dump(unsigned *out, unsigned *n_windows)
{
    if (hipMemcpyFromSymbol(n_windows, HIP_SYMBOL(g_row_dump_n), sizeof(unsigned)) != hipSuccess) return 1;
    const unsigned n = *n_windows < NR_ROW_DUMP_WINDOWS ? *n_windows : NR_ROW_DUMP_WINDOWS, zero = 0;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_row_dump), (size_t)n * 64 * sizeof(unsigned)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_row_dump_n), &zero, sizeof(unsigned)) == hipSuccess ? 0 : 1;
}
#else
#define NR_ROW_STAT(i, v) ((void)0)
#endif

// MODE: K6_FAST -- the default arithmetic, as described above -- or K6_EXACT / K6_EXACT_POW2 (NR_FLAG_EXACT_GRADIENT): the same
// groups of four records, every term with the reference's own operations: sum_c (I_c - ref_c) g_c rounded product by product
// (the colours, not the centred sums, are staged: 36 bytes per pixel, four workgroups per CU), `x * 2. / is` and `dist +- eps` in
// double, IEEE division, every sum in double -- a lane adds its terms to double accumulators, the matrix pipe adds the lanes.
// A masked lane divides 0 by 1 (the select sits in front of the division: no 0 / 0, no Inf * 0).
template <bool RGB, bool ALPHA, int MODE, bool CHUNKED>
__global__ __launch_bounds__(rowk::NT, MODE == K6_FAST ? 5 : 4) void k_bpm_row(
    const int32_t *__restrict__ fi_map, const float *__restrict__ rgb_map, const float *__restrict__ alpha_map,
    const float *__restrict__ g_rgb, const float *__restrict__ g_alpha, double *__restrict__ scratch,
    const int *__restrict__ band_lines, const int *__restrict__ band_start, const int *__restrict__ lines_ok,
    const BandLine *__restrict__ line_buf, size_t cap, int F, int S, int W, int CH, float eps_f, double eps_d, int B,
    uint4 *__restrict__ zero16, size_t n_zero16)
{
    using namespace rowk;
    constexpr bool EXACT = MODE != K6_FAST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
    // A workgroup takes a band of W lines and, of those lines, a CHUNK of CH pixels (all of them up to raster 512, half a line
    // above): the sweeps of a record are sums over pixels, so the part of a sweep inside each chunk can be summed by a
    // workgroup of its own -- every term is formed exactly as in an unchunked line (t = d1 - d1_cross from the pixel's position
    // along the whole line), the chunks' sums meet in the double atomics of the flush.  What it buys on rasters above 512:
    // two-line bands instead of one-line ones.  A band of COLUMNS is staged in pieces of W pixels per image row, and with
    // one-line bands 4 of every 64 bytes fetched were used: the column bands' staging was over half of the kernel at raster
    // 1024 (k_bpm_row with the bands of one axis only, without sweeps, 32 views: columns 545 us, rows 125).  What it costs: a
    // record is set up once per chunk its sweeps reach -- which is why the chunks are not shorter: quarter lines with four-line
    // bands stage the columns in 310 us instead of 545 and lose it all again (K6 stage, us, chunks of <= 1024 / 512 / 256 pixels:
    // 32 views at 1024^2 1015 / 960 / 1138, 4 views 204 / 181 / 206, 64 views at 512^2 512 / 512 / 547).
    // (CHUNKED = false: whole lines -- an instantiation of its own, because the chunk's offset as a run-time zero cost the
    // headline launch 1.4 %: K6 stage 166.0 -> 168.4 us)
    const unsigned n_bands = (unsigned)(S + W - 1) / (unsigned)W, n_ch = CHUNKED ? (unsigned)(S + CH - 1) / (unsigned)CH : 1u;
    const unsigned total_wg = n_bands * 2u * (unsigned)B * n_ch;
    const unsigned logical = xcd_block(total_wg);  // all bands of an image on one XCD (see k_bpm_fast)
    if (logical >= total_wg) return;
    if (n_zero16) {  // the fused backward's zero fill of grad_textures rides along (see k_bpm_fast)
        const size_t per = (n_zero16 + total_wg - 1) / total_wg, z_lo = (size_t)logical * per, z_hi = min(n_zero16, z_lo + per);
        for (size_t k = z_lo + tid; k < z_hi; k += NT) zero16[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    const unsigned bandidx = logical / n_ch;  // (the chunks of a band next to each other: they read the same records)
    const int c0 = CHUNKED ? (int)(logical - bandidx * n_ch) * CH : 0, SL = CHUNKED ? min(CH, S - c0) : S;  // the chunk: pixels [c0, c0 + SL) of the band's lines
    const int band = (int)(bandidx % n_bands), axis = (int)((bandidx / n_bands) & 1u), b = (int)(bandidx / (2u * n_bands));
    if (lines_ok[b] == 0) return;  // records beyond the buffer: k_bpm_fast's scan path serves this image
    const int band_lo = band * W, nld = min(W, S - band_lo);
    const size_t lt = ((size_t)b * 2 + axis) * S + band_lo;  // the band's lines in the per-line tables (band width 1)
    int n_tot = 0;
    for (int l = 0; l < nld; ++l) n_tot += band_lines[lt + l];
    if (n_tot == 0) return;  // no visible face has a line here

    constexpr int NC = RGB ? 4 : 1;  // floats per pixel in the gradient array: (alpha, r, g, b) or alpha alone
    // a line's chunk in LDS: SP pixels, a multiple of 32 (an even number of segments); the pixels behind the chunk hold zeros.
    // From here on pixel positions along a line are LOCAL: 0 is the chunk's first pixel.
    const int SP = (SL + 31) & ~31, nsl = SP / SEG;
    const unsigned npx = (unsigned)(W * SP), G_BYTES = (unsigned)g_bytes<RGB>(npx), C_BYTES = (unsigned)c_bytes<RGB, EXACT>(npx), FI_BYTES = npx * 4u;
    float *const s_g = (float *)smem;                                  // [W][SP][NC] gradients
    float *const s_p = (float *)(smem + G_BYTES);                      // [W][SP] sum_c (I_c - K_c) g_c; exact mode: [W][SP][NC] colours
    int *const s_fi = (int *)(smem + G_BYTES + C_BYTES);               // [W][SP] face index
    float4 *const s_k = (float4 *)(smem + G_BYTES + C_BYTES + FI_BYTES);  // [W] K of each line: (alpha, r, g, b) (not in the exact mode)
    unsigned char *wave_mem = smem + G_BYTES + C_BYTES + FI_BYTES + (EXACT ? 0u : (unsigned)K_BYTES) + (unsigned)wave * WAVE_BYTES;
    double2 *acc = (double2 *)wave_mem;  // [WIN] a record's two out-sweep sums (magnitudes) ...
    int *hist = (int *)wave_mem;         // ... after the sort's counters are done with the same bytes
    const int n_parts = max(1, NW / W);  // (a band narrower than the workgroup has waves: they share the windows of a line)
    const BandLine *recs_b = line_buf + (size_t)b * cap;
    const size_t img = (size_t)b * S * S;
    // map index of pixel d1 of band line ld (:587-593)
    auto map_index = [&](int ld, int d1) { return axis ? img + (size_t)(band_lo + ld) * S + (c0 + d1) : img + (size_t)(c0 + d1) * S + band_lo + ld; };
    // (alpha, r, g, b) of a pixel, straight from the maps
    auto map_colour = [&](size_t gi) {
        float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (!RGB || ALPHA) c.x = alpha_map[gi];
        if constexpr (RGB) { c.y = rgb_map[3 * gi]; c.z = rgb_map[3 * gi + 1]; c.w = rgb_map[3 * gi + 2]; }
        return c;
    };
    // K of a line: the colour of its first pixel (see above); not finite: 0
    auto line_k = [&](int ld) {
        if constexpr (EXACT) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float4 k = map_colour(map_index(ld, 0));
        const bool fin = fabsf(k.x) <= 3.0e38f && fabsf(k.y) <= 3.0e38f && fabsf(k.z) <= 3.0e38f && fabsf(k.w) <= 3.0e38f;
        if (!fin) k = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return k;
    };
    // (the first window of the wave's first line is requested in front of the staging loads: one global round trip less on the
    // workgroup's critical path)
    int4 hh_first = make_int4(1, 1, 0, 0);
    float4 qq_first = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (wave < nld * n_parts) {
        const int ld = wave / n_parts, part = wave - ld * n_parts;
        const int n_rec = band_lines[lt + ld], r = (part / window_team(n_rec, n_parts, nsl)) * WIN + lane;  // (shared windows: below)
        if (r < n_rec) {
            const BandLine *R = recs_b + band_start[lt + ld] + r;
            hh_first = *reinterpret_cast<const int4 *>(R);
            qq_first = *reinterpret_cast<const float4 *>(&R->cross);
        }
    }
    // ---- stage the band: gradients, sums, face indices
    {
        auto put = [&](int ld, int d1, int fi, const float4 &k, float al, float ga, float r, float g, float bl, float gr, float gg, float gb) {
            const int l = ld * SP + d1;
            s_fi[l] = fi;
            if constexpr (RGB) {
                *reinterpret_cast<float4 *>(s_g + 4 * (size_t)l) = make_float4(ga, gr, gg, gb);
                if constexpr (EXACT)
                    *reinterpret_cast<float4 *>(s_p + 4 * (size_t)l) = make_float4(al, r, g, bl);
                else
                    s_p[l] = (float)((ALPHA ? (double)(al - k.x) * (double)ga : 0.0) + (double)(r - k.y) * (double)gr +
                                     (double)(g - k.z) * (double)gg + (double)(bl - k.w) * (double)gb);
            } else {
                s_g[l] = ga;
                s_p[l] = EXACT ? al : (al - k.x) * ga;
            }
            if constexpr (!EXACT) { if (d1 == 0) s_k[ld] = k; }
        };
        // four adjacent pixels of a map row with one 16-byte load per field (see fast_stage)
        auto put_quad = [&](size_t g, int ld0, int d10, int ld_step, int d1_step) {
            const int4 vf = *reinterpret_cast<const int4 *>(fi_map + g);
            float4 va = make_float4(0, 0, 0, 0), vg = va;
            if (ALPHA) {
                va = *reinterpret_cast<const float4 *>(alpha_map + g);
                vg = *reinterpret_cast<const float4 *>(g_alpha + g);
            }
            float rr[12], qq[12];
#pragma unroll
            for (int k = 0; k < 12; k++) rr[k] = qq[k] = 0.0f;
            if (RGB) {
                const float4 *pr = reinterpret_cast<const float4 *>(rgb_map + 3 * g);
                const float4 *pg = reinterpret_cast<const float4 *>(g_rgb + 3 * g);
                const float4 r0 = pr[0], r1 = pr[1], r2 = pr[2], q0 = pg[0], q1 = pg[1], q2 = pg[2];
                rr[0] = r0.x; rr[1] = r0.y; rr[2] = r0.z; rr[3] = r0.w; rr[4] = r1.x; rr[5] = r1.y; rr[6] = r1.z; rr[7] = r1.w;
                rr[8] = r2.x; rr[9] = r2.y; rr[10] = r2.z; rr[11] = r2.w;
                qq[0] = q0.x; qq[1] = q0.y; qq[2] = q0.z; qq[3] = q0.w; qq[4] = q1.x; qq[5] = q1.y; qq[6] = q1.z; qq[7] = q1.w;
                qq[8] = q2.x; qq[9] = q2.y; qq[10] = q2.z; qq[11] = q2.w;
            }
            const int fis[4] = {vf.x, vf.y, vf.z, vf.w};
            const float als[4] = {va.x, va.y, va.z, va.w}, gas[4] = {vg.x, vg.y, vg.z, vg.w};
            float4 k = line_k(ld0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j > 0 && ld_step) k = line_k(ld0 + j * ld_step);
                put(ld0 + j * ld_step, d10 + j * d1_step, fis[j], k, als[j], gas[j], rr[3 * j], rr[3 * j + 1], rr[3 * j + 2], qq[3 * j],
                    qq[3 * j + 1], qq[3 * j + 2]);
            }
        };
        auto put_one = [&](int ld, int d1) {
            const size_t g = map_index(ld, d1);
            float al = 0, ga = 0, r = 0, gn = 0, bl = 0, gr = 0, gg = 0, gb = 0;
            if (ALPHA) { al = alpha_map[g]; ga = g_alpha[g]; }
            if (RGB) {
                r = rgb_map[3 * g]; gn = rgb_map[3 * g + 1]; bl = rgb_map[3 * g + 2];
                gr = g_rgb[3 * g]; gg = g_rgb[3 * g + 1]; gb = g_rgb[3 * g + 2];
            }
            put(ld, d1, fi_map[g], line_k(ld), al, ga, r, gn, bl, gr, gg, gb);
        };
        if (SP != SL) {  // the pixels behind the chunk: no gradient, nobody's
            const int pad = SP - SL;
            for (int i = tid; i < nld * pad; i += NT) {
                const int ld = i / pad, l = ld * SP + SL + (i - ld * pad);
                s_fi[l] = -1;
                if constexpr (EXACT && RGB) *reinterpret_cast<float4 *>(s_p + 4 * (size_t)l) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                else s_p[l] = 0.0f;
                if constexpr (RGB) *reinterpret_cast<float4 *>(s_g + 4 * (size_t)l) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                else s_g[l] = 0.0f;
            }
        }
        if (axis && (S & 3) == 0) {  // a band line is an image row: thread -> (line, quad of 4 consecutive pixels)
            const int quads = SL >> 2;  // (c0 and SL are multiples of 4 with S)
            for (int i = tid; i < nld * quads; i += NT) {
                const int ld = i / quads, x = (i - ld * quads) << 2;
                put_quad(img + (size_t)(band_lo + ld) * S + c0 + x, ld, x, 0, 1);
            }
        } else if (axis) {  // thread -> (line, x), x fastest (coalesced 4- and 12-byte loads)
            for (int i = tid; i < nld * SL; i += NT) put_one(i / SL, i % SL);
        } else if (nld == 4 && (S & 3) == 0 && (band_lo & 3) == 0) {  // 4 adjacent columns: one thread per row
            for (int y = tid; y < SL; y += NT) put_quad(img + (size_t)(c0 + y) * S + band_lo, 0, y, 1, 0);
        } else {  // generic columns: thread -> (row d1, line ld) with ld fastest
            for (int i = tid; i < nld * SL; i += NT) put_one(i % nld, i / nld);
        }
    }
    __syncthreads();

    float eps_v = eps_f;
    asm volatile("" : "+v"(eps_v));
    // the exact mode's term: :649-651 / :654-656 (x * 2. / S: an exact float scaling when S is a power of two); ct = c * t
    const unsigned eps_hi = (unsigned)__double2hiint(eps_d), eps_lo = (unsigned)__double2loint(eps_d);
    const double s_d = (double)S, two_over_s_d = 2.0 / (double)S;
    const float two_over_s_f = (float)(2.0 / (double)S);
    // `x * 2. / S` of a float x, rounded to float, without the double division when S is not a power of two: the double product
    // x * RN(2 / S) is within 2^-52 of the quotient, and so is the correctly rounded double quotient the reference's expression
    // forms; the quotient itself is at least 2^-36 away (relative) from every point where the rounding to float changes -- such
    // a point is M 2^f with M odd and of 25 bits, x = X 2^e with X below 2^24, S = 2^k s with s odd, >= 3 and below 2^11:
    // X 2^e = M s 2^(f+k) is impossible (M s is odd and has 26 bits or more), and two different multiples of 2^f differ by 2^f
    // at least, i.e. by 1 / (M S) > 2^-36 of the quotient -- so both doubles round to the same float.  The argument needs the
    // float's full 24 bits: a result below the normal range takes the division (c is a ratio of differences of pixel coordinates
    // and |t| a distance between them: such a product is 0 or far above 1e-38; the branch keeps the argument free of that).
    // k_bpm_fast keeps the division everywhere; tests/test_hip_parity.py compares the two kernels' exact modes bit for bit.
    auto exact_scale = [&](float ct) {
        if constexpr (MODE == K6_EXACT_POW2) return ct * two_over_s_f;
#ifdef NR_ROW_TRUE_DIV
        return (float)((double)ct * 2.0 / s_d);
#else
        float q = (float)((double)ct * two_over_s_d);
        if (__builtin_expect(!(fabsf(q) >= 1.17549435e-38f) && ct != 0.0f, 0)) q = (float)((double)ct * 2.0 / s_d);
        return q;
#endif
    };
    auto exact_dist = [&](float ct) {
        const float dist = exact_scale(ct);
        return (float)((double)dist + signed_eps(dist, eps_hi, eps_lo));  // + eps when 0 < dist, - eps otherwise
    };
    // The reciprocal of phase A's terms -- the in sweep and the out sweep's first pixel: the pixels next to the crossing point,
    // whose terms are a record's largest -- with one Newton step: v_rcp_f32's 1 ulp becomes ~0.5.  A full ulp on the largest
    // term of an entry that cancels down to the metric's floor (1e-3 of the largest gradient) reads as 2^-23 / 1e-3 = 1.2e-4:
    // round 5's soak found such a scene at 1.13e-4.  Two fused multiply-adds per term of a lane-per-record loop: not measurable;
    // on every visit of phase B they would be +25 % of the kernel's vector work.
    auto recip_n = [](float y) {
        const float r = __builtin_amdgcn_rcpf(y);
        return __builtin_fmaf(__builtin_fmaf(-y, r, 1.0f), r, r);
    };
    // sum_c (I_c - ref_c) g_c with the reference's operations in its order (:631-638 / :709-716)
    auto exact_diff = [&](const float4 &c4, const float4 &cr, const float4 &g4) {
        if constexpr (!RGB) return (c4.x - cr.x) * g4.x;
        float d = ALPHA ? (c4.x - cr.x) * g4.x + (c4.y - cr.y) * g4.y : (c4.y - cr.y) * g4.y;
        d += (c4.z - cr.z) * g4.z;
        d += (c4.w - cr.w) * g4.w;
        return d;
    };
    // the lane's block (record of a group) and its pixel inside a segment (see above)
    const int row = (lane >> 2) & 3, l16 = ((((lane >> 4) + ((lane >> 2) & 2)) & 3) << 2) | (lane & 3);
    for (int vt = wave; vt < nld * n_parts; vt += NW) {
        const int ld = vt / n_parts, part = vt - ld * n_parts;
        const int n_rec = band_lines[lt + ld];
        if (n_rec == 0) continue;
        const BandLine *recs = recs_b + band_start[lt + ld];
        const int base = ld * SP;
        float4 kc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (!EXACT) kc = s_k[ld];
        // A line with fewer windows than its band gives it waves (bands narrower than the workgroup: rasters above 256, small
        // launches; one or two windows of records per line are the rule on meshes of a few thousand faces): the waves form
        // TEAMS that share a window instead of idling -- every member repeats the lane-per-record set-up (the records, the
        // reference colours, the sort: the same result in each), member 0 walks the in sweeps and the out pixels, and the
        // groups of phase B are dealt out in turn; every member flushes what it summed (a record's in and out parts may then
        // arrive as two atomic additions instead of one).  The sums of a record are formed exactly as before, by whichever
        // wave takes its group.  (K6 stage, us, teams off / on: 4 views at 1024^2 235 / 204, 32 views 1165 / 1028, 64 views at
        // 768^2 1195 / 1154, at 512^2 508 / 500; the exact mode at 640^2 ... 1024^2 -15 ... -22 %.)
        const int team = window_team(n_rec, n_parts, nsl), member = part & (team - 1), n_teams = n_parts / team;
        const bool sweeps_mine = member == 0;
        const int w_first = (part / team) * WIN;
        for (int w0 = w_first; w0 < n_rec; w0 += WIN * n_teams) {
            const int nw = min(WIN, n_rec - w0);
            // ---- phase A: lane = record.  The lane keeps its record for the flush, walks the record's in sweep, and prepares what
            // its block will be handed in phase B: the reference colour of the OUT sweep (the in pixel, :594-601) minus K, |c0|
            // (carrying the direction in its sign bit), |c1|, -direction * crossing point, the number of segments of the sweep.
            int4 hh = hh_first;
            float4 qq = qq_first;
            if (!(vt == wave && w0 == w_first)) {  // (not the window requested in front of the staging)
                hh = make_int4(1, 1, 0, 0);
                qq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (lane < nw) {
                    const BandLine *R = recs + w0 + lane;
                    hh = *reinterpret_cast<const int4 *>(R);
                    qq = *reinterpret_cast<const float4 *>(&R->cross);
                }
            }
            // (local positions: the in pixel and the out pixel may lie outside the chunk.  The crossing point stays what it is: a
            // pixel's t = d1 - d1_cross is formed from its position along the WHOLE line, c0 + local -- the crossing point minus c0
            // would be rounded wherever c0 is the larger of the two)
            const int flags = (hh.z >> 24) & 0xff, d1_in = (hh.z & 0xffff) - c0;
            const int og_from = hh.y & 0xffff, og_to = hh.y >> 16;  // the out sweep along the whole line ...
            const int ig_from = hh.x & 0xffff, ig_to = hh.x >> 16;  // ... and the in sweep
            const bool has_out = og_from <= og_to;  // (:604: the in pixel is the face's)
            const bool has_in = ig_from <= ig_to;   // (every record whose in and out pixel lie inside the image)
            const bool dpos = (flags & 8) != 0;   // direction > 0: the sweep ends at the last pixel of the line, else it starts at pixel 0
            const int d1_out = d1_in + (dpos ? 1 : -1);
            // what of the record lies in this chunk: its in sweep's pixels, its out pixel (phase A's), and phase B's part of the
            // out sweep -- [out pixel + 1, end of the line] or [0, out pixel - 1]; the exact mode: with the out pixel
            const int in_from = max(ig_from, c0) - c0, in_to = min(ig_to, c0 + SL - 1) - c0;
            const bool out_px_here = has_out && d1_out >= 0 && d1_out < SL;
            const int bg_from = og_from + ((dpos && !EXACT) ? 1 : 0), bg_to = og_to - ((!dpos && !EXACT) ? 1 : 0);
            const int b_from = max(bg_from, c0) - c0, b_to = min(bg_to, c0 + SL - 1) - c0;
            const bool has_b = has_out && b_from <= b_to;
            const bool in_here = has_in && in_from <= in_to;
            // the two reference colours straight from the maps: the in pixel for the out sweep (:594-601), the out pixel for the
            // in sweep (:697-700); minus K for the two-sum form of the visits
            float4 c_in = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c_out = c_in;
            if (has_in && (has_b || in_here || out_px_here)) {  // (a record with a pixel in this chunk)
                c_in = map_colour(map_index(ld, d1_in));
                c_out = map_colour(map_index(ld, d1_out));
            }
            const float4 oref = make_float4(c_in.x - kc.x, c_in.y - kc.y, c_in.z - kc.z, c_in.w - kc.w);
            const float4 iref = make_float4(c_out.x - kc.x, c_out.y - kc.y, c_out.z - kc.z, c_out.w - kc.w);
            // The two pixels next to the crossing point carry a record's largest terms (|t| <= 1).  Their colours are the two
            // reference colours, so their colour differences are formed DIRECTLY here -- sum_c (I_c - ref_c) g_c, the reference's
            // own expression -- and only the pixels further out go through the centred sums P: the in pixel is the first pixel
            // of the in sweep (below), the out pixel the first one of the out sweep, which phase B then leaves out (td > 1).
            auto direct_diff = [&](const float4 &ci, const float4 &cr, int l) {
                float4 g4;
                if constexpr (RGB) g4 = lds_px4(s_g + 4 * (size_t)l);
                else g4 = make_float4(s_g[l], 0.0f, 0.0f, 0.0f);
                float d;
                if constexpr (EXACT) {
                    d = exact_diff(ci, cr, g4);
                } else if constexpr (RGB) {
                    d = ALPHA ? __builtin_fmaf(ci.y - cr.y, g4.y, (ci.x - cr.x) * g4.x) : (ci.y - cr.y) * g4.y;
                    d = __builtin_fmaf(ci.z - cr.z, g4.z, d);
                    d = __builtin_fmaf(ci.w - cr.w, g4.w, d);
                } else {
                    d = (ci.x - cr.x) * g4.x;
                }
                return d;
            };
            float f0 = 0.0f, f1 = 0.0f;  // the out pixel's terms (magnitudes: the sign goes on at the flush, with the out sweep's)
            if (out_px_here && !EXACT && sweeps_mine) {  // (the exact mode: phase B walks the whole out sweep)
                const float d = direct_diff(c_out, c_in, base + d1_out);                           // :631-638
                const float dm = !(d <= 0.0f) ? d : 0.0f;                                           // :647
                const float t = fabsf((float)(c0 + d1_out) - qq.x);
                f0 = dm * recip_n(__builtin_fmaf(fabsf(qq.y), t, eps_v));                          // :649-651
                f1 = dm * recip_n(__builtin_fmaf(fabsf(qq.z), t, eps_v));                          // :654-656
            }
            if constexpr (!EXACT) {
                // Phase B tells the out pixel from the rest of the sweep by |t| <= 1 -- exact for every pixel but one: with the
                // crossing point within half an ulp of the in pixel's centre (a vertex snapped onto a pixel centre, the cross
                // product a rounding below it) the SECOND pixel's |t| = 1 + 2^-24 rounds to 1 and phase B leaves it out as well.
                // It is taken here, by the same comparison on the same float, with the centred sums phase B would have used
                // (tests/test_fuzz_gpu.py, seed 118: the term of such a pixel was missing, 27 % of a corner face's gradient).
                const int d1_2 = d1_out + (dpos ? 1 : -1);
                const float t2 = fabsf((float)(c0 + d1_2) - qq.x);
                if (has_out && og_from < og_to && d1_2 >= 0 && d1_2 < SL && t2 <= 1.0f && sweeps_mine) {
                    const int l = base + d1_2;
                    float4 g4;
                    if constexpr (RGB) g4 = lds_px4(s_g + 4 * (size_t)l);
                    else g4 = make_float4(s_g[l], 0.0f, 0.0f, 0.0f);
                    float d = s_p[l];
                    if constexpr (!RGB || ALPHA) d = __builtin_fmaf(-oref.x, g4.x, d);
                    if constexpr (RGB) {
                        d = __builtin_fmaf(-oref.y, g4.y, d);
                        d = __builtin_fmaf(-oref.z, g4.z, d);
                        d = __builtin_fmaf(-oref.w, g4.w, d);
                    }
                    const float dm = !(d <= 0.0f) ? d : 0.0f;
                    f0 = __builtin_fmaf(dm, recip_n(__builtin_fmaf(fabsf(qq.y), t2, eps_v)), f0);
                    f1 = __builtin_fmaf(dm, recip_n(__builtin_fmaf(fabsf(qq.z), t2, eps_v)), f1);
                }
            }
            // segments of what phase B walks in this chunk: [b_from / 16, nsl) towards the end of the line, [0, b_to / 16] from pixel 0
            const int nseg = has_b ? (dpos ? nsl - (b_from >> 4) : (b_to >> 4) + 1) : 0;
            hist[lane] = 0;
            double in0 = 0.0, in1 = 0.0;
            if (in_here && !(NR_ROW_OFF & 2) && sweeps_mine) {
                const float cross = qq.x, c0k = qq.y, c1k = qq.z;
                const int fnr = __float_as_int(qq.w);
                const float d_first = (EXACT || d1_in < 0 || d1_in >= SL) ? 0.0f : direct_diff(c_in, c_out, base + d1_in);
                // (batches of IN_BATCH pixels whose LDS reads are requested together -- nine in-sweeps in ten are one batch;
                // IN_SEG float terms per double addition, like a piece of k_bpm_fast)
                for (int s0 = in_from; s0 <= in_to; s0 += IN_SEG) {
                    const int s1 = min(s0 + IN_SEG - 1, in_to);
                    float b0 = 0.0f, b1 = 0.0f;
                    for (int q0 = s0; q0 <= s1; q0 += IN_BATCH) {
                        int fi[IN_BATCH];
                        float4 g4[IN_BATCH], c4[IN_BATCH];
                        float p4[IN_BATCH];
#pragma unroll
                        for (int k = 0; k < IN_BATCH; ++k) {
                            const int l = base + min(q0 + k, s1);
                            fi[k] = s_fi[l];
                            c4[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            p4[k] = 0.0f;
                            if constexpr (EXACT) {
                                if constexpr (RGB) c4[k] = lds_px4(s_p + 4 * (size_t)l);
                                else c4[k].x = s_p[l];
                            } else {
                                p4[k] = s_p[l];
                            }
                            if constexpr (RGB) g4[k] = lds_px4(s_g + 4 * (size_t)l);
                            else g4[k] = make_float4(s_g[l], 0.0f, 0.0f, 0.0f);
                        }
#pragma unroll
                        for (int k = 0; k < IN_BATCH; ++k) {
                            float diff;                                                            // :709-716
                            if constexpr (EXACT) {
                                diff = exact_diff(c4[k], c_out, g4[k]);
                            } else {
                                diff = p4[k];
                                if constexpr (!RGB || ALPHA) diff = __builtin_fmaf(-iref.x, g4[k].x, diff);
                                if constexpr (RGB) {
                                    diff = __builtin_fmaf(-iref.y, g4[k].y, diff);
                                    diff = __builtin_fmaf(-iref.z, g4[k].z, diff);
                                    diff = __builtin_fmaf(-iref.w, g4[k].w, diff);
                                }
                                if (q0 + k == d1_in) diff = d_first;
                            }
                            // :707, :717 (a NaN diff goes through); a pixel beyond the batch's end repeats the last one: dropped
                            const bool take = (q0 + k <= s1) & (fi[k] == fnr) & !(diff <= 0.0f);
                            const float t = (float)(c0 + q0 + k) - cross;
                            if constexpr (EXACT) {
                                // (the select in front of the division: a pixel that is not taken divides 0 by 1)
                                const float dm = take ? diff : 0.0f;
                                const float y0 = take ? exact_dist(c0k * t) : 1.0f, y1 = take ? exact_dist(c1k * t) : 1.0f;
                                in0 -= (double)(dm / y0);                                            // :719-722
                                in1 -= (double)(dm / y1);                                            // :724-727
                            } else {
                                const float x0 = c0k * t, x1 = c1k * t;                               // :719 / :724 (2 / S folded into c)
                                const float y0 = x0 + ((0.0f < x0) ? eps_v : -eps_v);                 // :720-721 / :725-726
                                const float y1 = x1 + ((0.0f < x1) ? eps_v : -eps_v);
                                const float dm = take ? diff : 0.0f;
                                // (y is never 0 here: x and its eps have one sign, eps > 0 -- so 0 * (1 / y) adds nothing)
                                b0 = __builtin_fmaf(-dm, recip_n(y0), b0);                            // :722
                                b1 = __builtin_fmaf(-dm, recip_n(y1), b1);                            // :727
                            }
                        }
                    }
                    in0 += (double)b0;
                    in1 += (double)b1;
                }
            }
            // ---- the order of phase B: records by falling number of segments (counting sort: a counter per key in LDS; the
            // records without an out sweep behind the others, so that the positions are a permutation of the lanes, which one
            // ds_permute_b32 inverts: lane k then knows which lane holds the k-th record)
            wave_lds_handover();
            const int key = MAX_SEGS - nseg;  // (only of records phase B walks: 0 .. MAX_SEGS - 1)
            int rank = 0;
            if (has_b) rank = atomicAdd(&hist[key], 1);
            wave_lds_handover();
            const int cnt = hist[lane], incl = wave_incl_sum_dpp(cnt);
            const int n_out = __builtin_amdgcn_readlane(incl, 63);
            wave_lds_handover();
            hist[lane] = incl - cnt;
            wave_lds_handover();
            const unsigned long long no_out = __ballot(!has_b);
            int pos = n_out + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(no_out >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)no_out, 0u));
            if (has_b) pos = hist[key] + rank;
            const int inv = __builtin_amdgcn_ds_permute(pos << 2, lane);
            wave_lds_handover();
            NR_ROW_STAT(0, 1);        // windows
            NR_ROW_STAT(1, nw);       // records
            NR_ROW_STAT(2, n_out);    // records with an out sweep
#ifdef NR_ROW_STATS
            {
                unsigned wid = 0;
                if (lane == 0) wid = atomicAdd(&g_row_dump_n, 1u);
                wid = (unsigned)__builtin_amdgcn_readfirstlane((int)wid);
                if (wid < NR_ROW_DUMP_WINDOWS)
                    g_row_dump[(size_t)wid * 64 + lane] = lane < nw ? ((unsigned)nseg | (dpos ? 256u : 0u) | (has_b ? 512u : 0u) | 1024u) : 0u;
            }
            {
                int sn = nseg, si = has_b ? (b_to - b_from + 1) : 0;
                for (int o = 32; o > 0; o >>= 1) { sn += __shfl_xor(sn, o, WAVE); si += __shfl_xor(si, o, WAVE); }
                NR_ROW_STAT(3, sn);   // segments of the out sweeps
                NR_ROW_STAT(4, si);   // pixels of the out sweeps
            }
#endif
            // ---- phase B: four records at a time, one per block of 16 lanes
            // what a block is handed (from the lane that holds the record)
            const float v_ncd = dpos ? -qq.x : qq.x;
            // fast mode: |c0| carrying the direction in its sign bit, |c1|; exact mode: c0 * direction, c1 * direction (signed: a
            // product with +-1 is exact, so (c * direction) * (direction * t) is the reference's c * t bit for bit), and the
            // direction in the sign of the segment count
            const float v_ac0 = EXACT ? (dpos ? qq.y : -qq.y)
                                      : __uint_as_float((__float_as_uint(qq.y) & 0x7fffffffu) | (dpos ? 0u : 0x80000000u));
            const float v_ac1 = EXACT ? (dpos ? qq.z : -qq.z) : fabsf(qq.z);
            const int v_nseg = EXACT ? (dpos ? nseg : -nseg) : nseg;
            float4 oref_b = oref;  // the out sweep's reference colour as the blocks use it: minus K, or as it is (exact mode)
            if constexpr (EXACT) oref_b = c_in;
            const int g_first = 4 * member, g_step = 4 * team;  // (a shared window: every team-th group)
            int src_next = __builtin_amdgcn_ds_bpermute(((g_first + row) & 63) << 2, inv);
            for (int g0 = g_first; g0 < ((NR_ROW_OFF & 1) ? 0 : n_out); g0 += g_step) {
                const bool act = g0 + row < n_out;
                const int src = src_next;  // (of a block without a record: some lane; nothing of it is used)
                src_next = __builtin_amdgcn_ds_bpermute(((g0 + g_step + row) & 63) << 2, inv);
                const int sa = src << 2;
                const int r_nseg_s = __builtin_amdgcn_ds_bpermute(sa, v_nseg);
                const int r_nseg = EXACT ? abs(r_nseg_s) : r_nseg_s;
                float ncd = __int_as_float(__builtin_amdgcn_ds_bpermute(sa, __float_as_int(v_ncd)));
                const float ac0s = __int_as_float(__builtin_amdgcn_ds_bpermute(sa, __float_as_int(v_ac0)));
                const float ac1 = __int_as_float(__builtin_amdgcn_ds_bpermute(sa, __float_as_int(v_ac1)));
                float ra = 0.0f, rr = 0.0f, rg = 0.0f, rb = 0.0f;
                if constexpr (!RGB || ALPHA) ra = __int_as_float(__builtin_amdgcn_ds_bpermute(sa, __float_as_int(oref_b.x)));
                if constexpr (RGB) {
                    rr = __int_as_float(__builtin_amdgcn_ds_bpermute(sa, __float_as_int(oref_b.y)));
                    rg = __int_as_float(__builtin_amdgcn_ds_bpermute(sa, __float_as_int(oref_b.z)));
                    rb = __int_as_float(__builtin_amdgcn_ds_bpermute(sa, __float_as_int(oref_b.w)));
                }
                // (a block without a record: td = -Inf for every pixel, nothing is taken, nothing stored)
                if (!act) ncd = -__builtin_inff();
                const float sdir = EXACT ? (r_nseg_s < 0 ? -1.0f : 1.0f)
                                         : __uint_as_float(0x3f800000u | (__float_as_uint(ac0s) & 0x80000000u));
                // the group walks as many steps as its longest sweep has segments (the first block's: the order of the sort; the exact
                // mode, which walks in pairs, an even number of them); a sweep towards the end of the line ENDS with the group's last
                // step, one from pixel 0 starts with its first; pixels in front of a sweep or behind it are masked (td <= 0)
                const int steps = __builtin_amdgcn_readfirstlane(r_nseg), steps2 = EXACT ? (steps + 1) & ~1 : steps;
                NR_ROW_STAT(5, 1);       // groups
                NR_ROW_STAT(6, steps2);  // steps walked
                const bool rpos = EXACT ? r_nseg_s >= 0 : !(__float_as_uint(ac0s) >> 31);
                const int seg0 = rpos ? nsl - steps2 : 0;
                const int p0 = seg0 * SEG + l16;
                float pf = (float)(c0 + p0);  // (the position along the whole line: see the crossing point above)
                double A0 = 0.0, A1 = 0.0;
                if constexpr (EXACT) {
                    // ---- the exact mode: gradients and colours of a step (two 16-byte reads), the reference's term (rasterize.py
                    // :631-657) operation by operation, a double accumulator pair per lane; td = direction * (d1 - d1_cross) is the
                    // reference's t = d1 - d1_cross times +-1, and ac0s / ac1 are its c0 / c1 times the same +-1: the products
                    // c * t are the reference's bit for bit
                    const unsigned char *gp = (const unsigned char *)s_g + (size_t)(base + p0) * (NC * 4);
                    const unsigned char *cp = (const unsigned char *)s_p + (size_t)(base + p0) * (NC * 4);
                    const float4 cref = make_float4(ra, rr, rg, rb);
                    auto load_x = [&](int k, float4 &g4, float4 &c4) {
                        if constexpr (RGB) {
                            g4 = *reinterpret_cast<const float4 *>(gp + k * (SEG * NC * 4));
                            c4 = *reinterpret_cast<const float4 *>(cp + k * (SEG * NC * 4));
                        } else {
                            g4 = make_float4(*reinterpret_cast<const float *>(gp + k * (SEG * NC * 4)), 0.0f, 0.0f, 0.0f);
                            c4 = make_float4(*reinterpret_cast<const float *>(cp + k * (SEG * NC * 4)), 0.0f, 0.0f, 0.0f);
                        }
                    };
                    auto visit_x = [&](const float4 &g4, const float4 &c4, const float pfv) {
                        if constexpr (RGB && !ALPHA) asm volatile("" : : "v"(g4.x), "v"(c4.x));
                        const float d = exact_diff(c4, cref, g4);
                        const float td = __builtin_fmaf(sdir, pfv, ncd);    // > 0 exactly on the sweep's pixels
                        const bool keep = !(td <= 0.0f) && !(d <= 0.0f);   // (:647: a NaN diff goes through)
                        // (both distances are formed for every lane and the select sits in front of the division: a masked lane
                        // divides 0 by 1 -- and the loop has no branch)
                        const float e0 = exact_dist(ac0s * td), e1 = exact_dist(ac1 * td);
                        const float dm = keep ? d : 0.0f, y0 = keep ? e0 : 1.0f, y1 = keep ? e1 : 1.0f;
                        A0 -= (double)(dm / y0);                                                    // :649-651
                        A1 -= (double)(dm / y1);                                                    // :654-656
                    };
                    for (int s = 0; s < steps2; s += 2) {  // (two steps per iteration: their reads and their chains overlap)
                        float4 gA, cA, gB, cB;
                        load_x(0, gA, cA);
                        load_x(1, gB, cB);
                        visit_x(gA, cA, pf);
                        visit_x(gB, cB, pf + (float)SEG);
                        gp += 2 * SEG * NC * 4;
                        cp += 2 * SEG * NC * 4;
                        pf += (float)(2 * SEG);
                    }
                    A0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, 1.0, 0.0, 0, 0, 0);
                    A1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, 1.0, 0.0, 0, 0, 0);
                    A0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, 1.0, 0.0, 0, 0, 0);
                    A1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, 1.0, 0.0, 0, 0, 0);
                } else {
                    const unsigned char *gp = (const unsigned char *)s_g + (size_t)(base + p0) * (NC * 4);
                    const float *pp = s_p + base + p0;
                    auto visit = [&](const float4 &g4, const float pv, const float pfv, float &a0, float &a1) {
                        float d = pv;                                                                  // :631-638
                        if constexpr (!RGB || ALPHA) d = __builtin_fmaf(-ra, g4.x, d);
                        if constexpr (RGB) {
                            d = __builtin_fmaf(-rr, g4.y, d);
                            d = __builtin_fmaf(-rg, g4.z, d);
                            d = __builtin_fmaf(-rb, g4.w, d);
                        }
                        // direction * (d1 - d1_cross): > 0 exactly on the sweep's pixels, in (0, 1] on its first one -- phase A's
                        const float td = __builtin_fmaf(sdir, pfv, ncd);
                        const bool keep = !(td <= 1.0f) && !(d <= 0.0f);  // (:647: a NaN diff goes through)
                        const float dm = keep ? d : 0.0f;
                        const float y0 = __builtin_fmaf(fabsf(ac0s), fabsf(td), eps_v), y1 = __builtin_fmaf(ac1, fabsf(td), eps_v);  // :649-650 / :654-655
                        a0 = __builtin_fmaf(dm, __builtin_amdgcn_rcpf(y0), a0);                        // :651 (sign: the flush)
                        a1 = __builtin_fmaf(dm, __builtin_amdgcn_rcpf(y1), a1);                        // :656
                    };
                    // (plain 16-byte reads: lds_px4's barrier would make every read wait for its data on the spot; the instance without
                    // alpha keeps its first component formally alive BEHIND the visit instead, see lds_px4)
                    auto load = [&](int k, float4 &g4, float &pv) {
                        if constexpr (RGB) g4 = *reinterpret_cast<const float4 *>(gp + k * (SEG * NC * 4));
                        else g4 = make_float4(*reinterpret_cast<const float *>(gp + k * (SEG * NC * 4)), 0.0f, 0.0f, 0.0f);
                        pv = pp[k * SEG];
                    };
                    auto used = [&](const float4 &g4) {
                        if constexpr (RGB && !ALPHA) asm volatile("" : : "v"(g4.x));
                    };
                    // A lane adds the terms of its even and of its odd segments in float (<= 8 terms each at raster 256, 32 at 1024: the
                    // terms fall off like 1 / t); everything above is double: the matrix pipe adds the 2 x 16 sums of a block.  Steps
                    // alternate between the two sums, so a sweep's terms are always split into its even and its odd segments -- which
                    // of the two registers holds which depends on the group (seg0 may be odd), the two sets and their order do not, and
                    // the two sums meet in a double addition: a record's sums are the same bits whatever window, group or launch it is
                    // part of.  (Chains cut every 16 / 8 / 4 steps,
                    // each with a reduction of its own: +0 / +4 / +12 % kernel time at raster 256, error levels unchanged.)
                    float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
                    {
                        // Two steps per pair, two pairs of registers: while one pair is visited the other pair's reads are in flight
                        // (the scheduling barriers keep the compiler from gathering the reads at the top of the loop, where every
                        // iteration would wait for them).  Behind the group's last step the reads fetch what is never used (inside
                        // the workgroup's LDS: at most two segments past a line).
                        float4 gA, gB, gC, gD;
                        float pA, pB, pC, pD;
                        load(0, gA, pA);
                        load(1, gB, pB);
                        int s = 0;
                        for (; s + 4 <= steps2; s += 4) {
                            load(2, gC, pC);
                            load(3, gD, pD);
                            __builtin_amdgcn_sched_barrier(0);
                            visit(gA, pA, pf, a0, a1);
                            used(gA);
                            visit(gB, pB, pf + (float)SEG, b0, b1);
                            used(gB);
                            __builtin_amdgcn_sched_barrier(0);
                            load(4, gA, pA);
                            load(5, gB, pB);
                            __builtin_amdgcn_sched_barrier(0);
                            visit(gC, pC, pf + (float)(2 * SEG), a0, a1);
                            used(gC);
                            visit(gD, pD, pf + (float)(3 * SEG), b0, b1);
                            used(gD);
                            __builtin_amdgcn_sched_barrier(0);
                            gp += 4 * (SEG * NC * 4);
                            pp += 4 * SEG;
                            pf += (float)(4 * SEG);
                        }
                        if (const int rem = steps2 - s) {  // (the last one to three steps: two of them already requested)
                            if (rem == 3) load(2, gC, pC);
                            visit(gA, pA, pf, a0, a1);
                            used(gA);
                            if (rem >= 2) {
                                visit(gB, pB, pf + (float)SEG, b0, b1);
                                used(gB);
                            }
                            if (rem == 3) {
                                visit(gC, pC, pf + (float)(2 * SEG), a0, a1);
                                used(gC);
                            }
                        }
                    }
                    A0 = __builtin_amdgcn_mfma_f64_4x4x4f64((double)a0, 1.0, 0.0, 0, 0, 0);
                    A1 = __builtin_amdgcn_mfma_f64_4x4x4f64((double)a1, 1.0, 0.0, 0, 0, 0);
                    A0 = __builtin_amdgcn_mfma_f64_4x4x4f64((double)b0, 1.0, A0, 0, 0, 0);
                    A1 = __builtin_amdgcn_mfma_f64_4x4x4f64((double)b1, 1.0, A1, 0, 0, 0);
                    A0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A0, 1.0, 0.0, 0, 0, 0);
                    A1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A1, 1.0, 0.0, 0, 0, 0);
                }
                if (act && lane == (row << 2)) acc[src] = make_double2(A0, A1);  // (every lane of the block holds the sums)
            }
            wave_lds_handover();
            // ---- flush: in sweep + out sweep of each record -> global double scratch [list position][vertex][x|y].
            // Out sweep: grad -= diff / (sigma * |dist|), sigma = sign(c) * sign(t), sign(t) = the direction (t = d1 - d1_cross
            // keeps its sign beyond the crossing point): the magnitudes were summed, the sign goes on here.  :648 / :653 (and
            // :718 / :723): a contribution whose vertex sits on the line is not taken (its coefficient was Inf / NaN).
            if (lane < nw) {
                double2 a = make_double2(0.0, 0.0);
                if (has_b && ((pos >> 2) & (team - 1)) == member) a = acc[lane];  // (the wave that walked the record's group)
                a.x += (double)f0;
                a.y += (double)f1;
                const bool tneg = !dpos;
                // (the exact mode's sums carry their signs: -(diff / dist) term by term)
                const bool neg0 = EXACT || (((__float_as_uint(qq.y) >> 31) != 0) != tneg);
                const bool neg1 = EXACT || (((__float_as_uint(qq.z) >> 31) != 0) != tneg);
                const double t0 = (flags & 2) ? in0 + (neg0 ? a.x : -a.x) : 0.0;
                const double t1 = (flags & 4) ? in1 + (neg1 ? a.y : -a.y) : 0.0;
                const int tgt = hh.w, lpos = tgt & 0x0fffffff;
                double *dst = scratch + ((size_t)b * F + lpos) * 6 + (1 - axis);
                if (NR_ROW_OFF & 4) {
                    asm volatile("" : : "v"(t0), "v"(t1), "v"(dst));
                } else {
                    if (t0 != 0.0) atomicAdd(dst + 2 * ((tgt >> 28) & 3), t0);
                    if (t1 != 0.0) atomicAdd(dst + 2 * ((tgt >> 30) & 3), t1);
                }
            }
            wave_lds_handover();
        }
    }
}

// add: grad_faces holds what K8 left for the face (zeros from the compaction, then the gather's sums: the fused backward whose
// gather runs beside the line setup) and K6's rounded sums go on top -- the one float addition per element that the in-gather
// finish makes, operands exchanged
__global__ __launch_bounds__(256) void k_bpm_finalize(const double *__restrict__ scratch, const int *__restrict__ slot_of,
                                                      float *__restrict__ grad_faces, int F, int n_faces_total, int add)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_faces_total) return;
    const int pos = slot_of[i];
    float *o = grad_faces + (size_t)i * 9;
    if (add) {
        if (pos < 0) return;
        const double *src = scratch + ((size_t)(i / F) * F + pos) * 6;
#pragma unroll
        for (int v = 0; v < 3; v++) {
            o[3 * v + 0] = (float)src[2 * v + 0] + o[3 * v + 0];
            o[3 * v + 1] = (float)src[2 * v + 1] + o[3 * v + 1];
        }
        return;
    }
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (pos >= 0) {
        const double *src = scratch + ((size_t)(i / F) * F + pos) * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) a[k] = src[k];
    }
#pragma unroll
    for (int v = 0; v < 3; v++) {
        o[3 * v + 0] = (float)a[2 * v + 0];
        o[3 * v + 1] = (float)a[2 * v + 1];
        o[3 * v + 2] = 0.0f;  // K6 never touches z
    }
}

// ====================================================================================================

struct BpmLayout {
    size_t flags_off, scratch_off, count_off, chunk_off, cband_off, list_off, rng_off, slot_off, band_off, start_off, cursor_off, ok_off,
        lines_off, total, cap;
    int n_chunks;
};

// capacity (line records per image) of the band line buffer: a mesh has a few lines per face (teapot 3.8, dense meshes ~2);
// scenes of few large faces are covered by the S term.  Images beyond it take the scan path of k_bpm_fast.
size_t line_capacity(int F, int S)
{
    // + S sqrt(F): a mesh whose V visible faces tile a fixed share of the image has ~ sqrt(V) S records (teapot views: ~100 k at
    // 1024^2, over 8 F + 32 S from raster ~600 on: the scan path cost k_bpm_fast 30 % there and the lane-parallel kernel -- whose
    // overflow launch strides over one-line bands -- a factor).  1.2 S sqrt(F): a mesh that FILLS the image has ~2.1 sqrt(F) S
    // records, which this holds for every raster up to 1264 -- so up to k_bpm_row's largest raster nothing about a mesh's
    // coverage has to be guessed when the band kernel is chosen
    const size_t want = (size_t)8 * F + (size_t)32 * S + (size_t)(1.2 * (double)S * sqrt((double)F));
    // (the images' buffers follow one another: a stride near a multiple of 2 MiB -- 65 579 records x 32 B at the headline shape --
    // puts the same band of every image on the same memory channels and cost k_line_setup 3 us of 25; the stride is kept at
    // 34 KiB past a multiple of 64 KiB)
    return (want + 2047 - 1088) / 2048 * 2048 + 1088;
}

BpmLayout bpm_layout(int B, int F, int S)
{
    BpmLayout L;
    const size_t n = (size_t)B * F;
    L.flags_off = 0;                                  // n bytes (only used when the caller has no visible_faces)
    L.scratch_off = align_up(n, 256);                 // n * 6 doubles, indexed by list position
    L.count_off = align_up(L.scratch_off + n * 6 * sizeof(double), 256);
    L.n_chunks = (F + VIS_CHUNK - 1) / VIS_CHUNK;
    L.chunk_off = L.count_off + align_up((size_t)B * sizeof(int), 256);
    // rows of k_compact_par: [B][n_chunks][2 * n_bands], n_bands <= S
    L.cband_off = L.chunk_off + align_up((size_t)B * L.n_chunks * sizeof(int), 256);
    L.list_off = L.cband_off + (L.n_chunks <= SMALL_CHUNKS ? align_up((size_t)B * L.n_chunks * 2 * S * sizeof(int), 256) : 0);
    L.rng_off = L.list_off + align_up(n * sizeof(int), 256);
    L.slot_off = L.rng_off + align_up(n * 6 * sizeof(unsigned), 256);  // rng: [B][axis][position][edge]
    const size_t per_band = align_up((size_t)B * 2 * S * sizeof(int), 256);  // per (image, axis, band): at most S bands (W = 1)
    L.band_off = L.slot_off + align_up(n * sizeof(int), 256);
    L.start_off = L.band_off + per_band;
    L.cursor_off = L.start_off + per_band;
    L.ok_off = L.cursor_off + per_band;
    L.lines_off = L.ok_off + align_up((size_t)B * sizeof(int), 256);
    L.cap = line_capacity(F, S);
    L.total = L.lines_off + (size_t)B * L.cap * sizeof(BandLine);
    return L;
}

// the depth-only backward keeps just the visible-face lists (k_list_visible): [B] counts, then [B][F] list entries
struct ListsLayout {
    size_t count_off, list_off, total;
};
ListsLayout lists_layout(int B, int F)
{
    ListsLayout L;
    L.count_off = 0;
    L.list_off = align_up((size_t)B * sizeof(int), 256);
    L.total = L.list_off + (size_t)B * F * sizeof(int);
    return L;
}

// Shape of a band workgroup, chosen per launch by the raster size (round 4, scripts/k6_variants.py -> profiles/r04_k6_shapes.jsonl;
// K6 stage in us, variants timed in one process behind a throw-away pass: 512 threads / 4-line bands / 53 KB  ->  256
// threads / 2-line bands / 40 KB):
//   teapot 64 x 256^2  209 -> 204    16 x 256^2  99 -> 87    8 x 256^2  69 -> 65    64 x 384^2  499 -> 400   256 x 128^2  330 -> 273
//   config 4 (64 spiky meshes x 10 240 faces, 256^2)  433 -> 367
//   teapot 64 x 448^2  524 -> 579    64 x 512^2  682 -> 701    4 x 1024^2  338 -> 349
// i.e. while a two-line band leaves room for a window of >= 128 lines (raster <= 400 with colours) it wins, walked by four
// waves: the same ~110 lines and ~1 200 pieces per window as four lines give eight waves, in workgroups half the size --
// four per CU, a finer grain for the tail and for the mix of busy and empty bands.  Beyond that the wide shape stays.
struct BandShape {
    int threads;     // 256 | 512
    int w_max;       // widest band (lines)
    size_t budget;   // LDS per workgroup
};
BandShape band_shape(int S)
{
    if (S <= k6::SMALL_RASTER_MAX) return {256, 2, (size_t)40 * 1024};  // four workgroups per 160 KB CU
    // rasters of ~600 ... 830: a two-line band no longer fits 53 KB beside a useful line window, and a one-line band stages single
    // columns (64 teapot views, K6 stage: 576^2 857 us -> 608^2 1242, 640^2 1461); with 80 KB -- two workgroups per CU, still four
    // waves per SIMD -- the band stays two lines wide: 640^2 1061, 768^2 1574 -> 1400 (at 512^2 / 576^2 / 1024^2 the same budget
    // costs 5 ... 15 %: there the band width does not change)
    if (S > k6::WIDE_BUDGET_FROM && S <= k6::WIDE_BUDGET_TO) return {512, k6::WMAX, (size_t)80 * 1024};
    return {512, k6::WMAX, k6::LDS_BUDGET};
}

// LDS of k_bpm_fast: pixel arrays [W][SP] (face index 4 B, gradients and colours 16 B each -- 4 B each for alpha alone),
// coverage bits, and what is left is split between the line window (32 B record + two double sums per line; at most one line
// per thread) and the piece queue (2 or 4 B per descriptor).  Returns W (0: the raster does not fit, global fallback).
int fast_band_config(int S, bool rgb, const BandShape &shape, int w_max, size_t *lds_bytes, int *win, int *qcap)
{
    const size_t per_px = rgb ? 36 : 12, SP = (size_t)S + 4, dsz = S > 63 * FSEG ? 4 : 2;
    // pieces per line the split is made for: measured, stage times in us, raster 256: S/22 (192 lines, 3072 descriptors)
    // 240, S/32 (224, 2560) 230, S/48 (256, 2048: two rounds per window) 267; raster 512: S/22 (160, 4096) 906, S/32 (192,
    // 3584) 891, S/48 (224, 2560) 766 -- a band of a 512 x 512 teapot view has ~185 lines: one window instead of two
    const size_t segs_per_line = (size_t)S / (S <= 320 ? 32 : 48) + 3;
    const size_t NT = (size_t)shape.threads, LDS_BUDGET = shape.budget;
    const int win_max = BAND_WIN < shape.threads ? BAND_WIN : shape.threads;
    auto lines_bytes = [&](int ww) { return (sizeof(BandLine) + 16) * (size_t)ww; };
    for (int W = w_max; W >= 1; W >>= 1) {
        const size_t px = (size_t)W * SP * per_px + (size_t)W * (((SP + 31) / 32 + 3) / 4 * 4) * 4 + 16 /* bg */ + 32 /* spans */ + 64 /* scan */ +
                          8 * 16 /* alignment slack */;
        // the budget of the shape; a one-line band may take the whole LDS
        const size_t lim = (W == 1 && px + lines_bytes(32) + NT * dsz > LDS_BUDGET) ? 160 * 1024 : LDS_BUDGET;
        if (px + lines_bytes(32) + NT * dsz > lim) continue;
        const size_t rest = lim - px;
        int w = (int)(rest / (48 + segs_per_line * dsz));
        w = max(32, min(win_max, w / 32 * 32));
        while (w > 32 && lines_bytes(w) + NT * dsz > rest) w -= 32;
        if (W > 1 && w < 96) continue;  // a band this wide leaves no room for a useful line window: narrower bands
        size_t q = (rest - lines_bytes(w)) / dsz / NT * NT;
        q = q > 16384 ? 16384 : q;
        *win = w;
        *qcap = (int)q;
        *lds_bytes = px + lines_bytes(w) + q * dsz;
        return W;
    }
    return 0;
}

// A kernel that asks for more than 48 KB of dynamic LDS has to be told so (hipFuncSetAttribute).  The attribute is sticky,
// so the largest size granted so far is remembered per kernel instantiation and device and the launch path only makes the
// runtime call when a launch needs more than that -- in a steady loop, never.  (A cache of an idempotent driver setting,
// not library state: losing it would only repeat the call.)
struct LdsLimit {
    std::atomic<size_t> granted[32];
    std::mutex mtx;
    LdsLimit() { for (auto &g : granted) g.store(48 * 1024); }
    int ensure(const void *kern, size_t lds)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 31;
        if (lds <= granted[dev].load(std::memory_order_acquire) && dev != 31) return 0;
        // hipFuncSetAttribute SETS the limit: concurrent callers must never lower it, so the call and the record are one
        // critical section and the value passed is the maximum of what was granted and what is asked for
        std::lock_guard<std::mutex> lock(mtx);
        const size_t have = granted[dev].load(std::memory_order_relaxed);
        if (lds <= have && dev != 31) return 0;
        const size_t want = lds > have ? lds : have;
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e != 0) return e;
        granted[dev].store(want, std::memory_order_release);
        return 0;
    }
};

template <bool RGB, bool ALPHA, int MODE, int NT, bool OVF = false>
int launch_fast(const float *faces, const int32_t *fi, const float *rgb, const float *alpha, const float *g_rgb,
                const float *g_alpha, const int *vis_list, const int *vis_count, const unsigned *rng, double *scratch,
                const int *band_lines, const int *band_start, const int *lines_ok, const BandLine *line_buf, size_t cap,
                int B, int F, int S, int W, size_t lds, double eps, float k2s, int win_lines, int qcap, hipStream_t st,
                void *zero_ptr, size_t zero_bytes)
{
    static LdsLimit limit;  // one per instantiation
    constexpr int overflow_only = OVF ? 1 : 0;
    auto kern = k_bpm_fast<RGB, ALPHA, MODE, NT, OVF>;
    if (int rc = limit.ensure((const void *)kern, lds)) return rc;
    const unsigned total_wg = (unsigned)((S + W - 1) / W) * 2u * (unsigned)B;
    // 1-D grid: the kernel maps ids to (image, axis, band) per XCD
    // (overflow-only launch behind k_bpm_row: a resident grid that strides over the bands, k6::OVF_GRID workgroups.  With nothing
    // to do it costs 4.6 us in a step whatever the grid -- a launch (1.3 us for up to 256 workgroups that leave at once,
    // scripts/dev/empty_launch_probe.hip) and one dependent load of the images' verdicts from memory the line setup wrote with
    // atomics; with every image over the line buffer -- 32 teapot views at 1024^2 -- 1024 workgroups took 5.0 ms against 7.3 at 256)
    const unsigned grid = overflow_only ? (total_wg < k6::OVF_GRID ? total_wg : k6::OVF_GRID) : xcd_grid(total_wg);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, faces, fi, rgb, alpha, g_rgb, g_alpha,
                       vis_list, vis_count, rng, scratch, band_lines, band_start, lines_ok, line_buf, cap, F, S, W, S + 4,
                       eps, k2s, B, win_lines, qcap, (uint4 *)zero_ptr, zero_bytes / 16);
    return 0;
}

// k_bpm_row's chunks: a line is cut into ceil(S / 512) pieces of equal length (a multiple of 32 pixels, the last one shorter):
// the whole line up to raster 512, 2 x 320 at 640, 2 x 512 at 1024 (see the kernel) -- on meshes of fewer than 2^15 faces.  A
// record is set up once per chunk, and on a dense mesh (hundreds of records per line) that costs more than the column bands'
// staging gains: config 5 (655 360 faces at 1024^2) 1.34 ms per step with whole lines, 1.38 with half lines.
#ifndef NR_ROW_CHUNK_PX  // (development: the longest chunk)
#define NR_ROW_CHUNK_PX 512
#endif
int row_chunk(int S, int F)
{
    if (F >= (1 << 15)) return (S + 31) & ~31;
    const int n_ch = (S + NR_ROW_CHUNK_PX - 1) / NR_ROW_CHUNK_PX, len = (S + n_ch - 1) / n_ch;
    return (len + 31) & ~31;
}

// k_bpm_row's band: the widest power of two of lines (<= one per wave) whose chunks fit its LDS regions (rowk::MAX_PX); 0: the
// raster is too large for it (k_bpm_fast takes the launch)
int row_band_config(int S, int F, bool rgb, bool exact, int B, size_t *lds_bytes)
{
    if ((((size_t)S + 31) & ~(size_t)31) / rowk::SEG > (size_t)rowk::MAX_SEGS) return 0;
    const size_t SP = (size_t)row_chunk(S, F), n_ch = ((size_t)S + SP - 1) / SP;
    for (int W = rowk::NW; W >= 1; W >>= 1) {
        // (small launches: narrower bands while there are few band workgroups -- the waves of a workgroup then share the windows of
        // a line.  One-line bands stage single columns and most lines have one window, i.e. one busy wave: only below
        // ROW_MIN_WGS_1 workgroups.  K6 stage, us, four- / two- / one-line bands: 4 views 46.3 (1) -> 45.4 (2); 8 views 62.8 (4),
        // 59.3 (2), 61.8 (1); 16 views 79.9 (4), 75.6 (2), 86.0 (1); 8 views at 512^2 115.6 (2), 122.5 (1))
        const size_t wgs = (size_t)B * 2 * ((S + W - 1) / W) * n_ch;
        if ((W > 2 && wgs < k6::ROW_MIN_WGS) || (W == 2 && wgs < k6::ROW_MIN_WGS_1)) continue;
        if ((size_t)W * SP > (size_t)rowk::MAX_PX) continue;
        const size_t npx = (size_t)W * SP;
        *lds_bytes = rgb ? (exact ? rowk::lds_bytes<true, true>(npx) : rowk::lds_bytes<true, false>(npx))
                         : (exact ? rowk::lds_bytes<false, true>(npx) : rowk::lds_bytes<false, false>(npx));
        return W;
    }
    return 0;
}

template <bool RGB, bool ALPHA, int MODE>
int launch_row(const int32_t *fi, const float *rgb, const float *alpha, const float *g_rgb, const float *g_alpha, double *scratch,
               const int *band_lines, const int *band_start, const int *lines_ok, const BandLine *line_buf, size_t cap, int B,
               int F, int S, int W, size_t lds, double eps, hipStream_t st, void *zero_ptr, size_t zero_bytes)
{
    const int CH = row_chunk(S, F);
    const unsigned n_ch = (unsigned)((S + CH - 1) / CH), total_wg = (unsigned)((S + W - 1) / W) * 2u * (unsigned)B * n_ch;
    auto go = [&](auto chunked) {
        constexpr bool C = decltype(chunked)::value;
        static LdsLimit limit;  // one per instantiation (40 KB at most with the product's constants: never raised)
        if (int rc = limit.ensure((const void *)k_bpm_row<RGB, ALPHA, MODE, C>, lds)) return rc;
        hipLaunchKernelGGL((k_bpm_row<RGB, ALPHA, MODE, C>), dim3(xcd_grid(total_wg)), dim3(rowk::NT), lds, st, fi, rgb, alpha, g_rgb,
                           g_alpha, scratch, band_lines, band_start, lines_ok, line_buf, cap, F, S, W, CH, (float)eps, eps, B,
                           (uint4 *)zero_ptr, zero_bytes / 16);
        return 0;
    };
    return n_ch > 1 ? go(std::true_type()) : go(std::false_type());
}

}  // namespace

NR_API size_t nr_backward_workspace_bytes(int32_t B, int32_t F, int32_t S, int32_t return_rgb, int32_t return_alpha)
{
    if (check_sizes(B, F, S)) return 0;
    // depth only (no K6): nothing but the per-image lists of the faces that own a pixel, for the K8 gather
    if (!return_rgb && !return_alpha) return lists_layout(B, F).total;
    return bpm_layout(B, F, S).total;
}

// Which band kernel a call takes: the band width (lines per workgroup) of k_bpm_row and its LDS bytes, or 0 for k_bpm_fast.
// Both arithmetic modes have ONE band kernel since round 6, k_bpm_row: it is ahead of k_bpm_fast on every shape measured
// (profiles/r06_k6_kernels.md: 8 ... 128 teapot views at 256^2, 64 views at rasters 320 ... 768, 32 and 4 views at 1024^2, 256
// views at 128^2, 1024 at 32^2, configs 4 and 5; the exact mode: 64 views at 256^2 and 512^2), so nothing about the call's size
// enters the choice -- a batch and its shards take the same kernel.  k_bpm_fast keeps what k_bpm_row does not do: the in-kernel
// face scan (NR_FLAG_K6_SCAN, and the images of a call whose records exceed the line buffer: an overflow-only launch behind
// k_bpm_row), rasters beyond k_bpm_row's LDS band (> 1024), the default mode with eps = 0 (a lane outside a sweep multiplies 0
// by 1 / (|c t| + eps), and t = 0 -- the crossing point on a pixel centre -- would make that 0 * Inf; the exact mode selects in
// front of its division and takes any eps), and NR_FLAG_K6_LEGACY (tests, measurements).
// With k_bpm_row the band tables and the line records are binned per LINE (band width 1).
int k6_row_band(int B, int F, int S, bool rgb, double eps, int flags, bool fast_fits, size_t *row_lds)
{
    const bool exact = (flags & NR_FLAG_EXACT_GRADIENT) != 0;
    const bool possible = !(flags & (NR_FLAG_K6_SCAN | NR_FLAG_K6_LEGACY | NR_FLAG_K6_GLOBAL)) && B <= 65535 && fast_fits &&
                          (exact || (float)eps >= 1e-30f);
    return possible ? row_band_config(S, F, rgb, exact, B, row_lds) : 0;
}

// Measurement hook (include/nr_hip_profile.h, nr_profile_band_kernel): a pair of events around the band kernel's launch.  Only in the
// measurement build of the library (libnr_hip_prof.so, -DNR_PROFILE_HOOK: neural_renderer_amd._build.build_profile); the product
// library keeps no such state and does not export the two functions.
#ifdef NR_PROFILE_HOOK
#include "../../include/nr_hip_profile.h"
namespace {
struct BandKernelTimer {
    bool on = false, recorded = false;
    int which = -1;  // the bracketed launch: 0 k_bpm_fast, 1 k_bpm_row
    hipEvent_t start = nullptr, stop = nullptr;
} g_band_timer;
}  // namespace

NR_API int nr_profile_band_kernel(int32_t enable)
{
    BandKernelTimer &t = g_band_timer;
    if (enable && !t.start) {
        if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess) {
            t.start = t.stop = nullptr;
            return launch_status();
        }
    }
    t.on = enable != 0 && t.start != nullptr;
    t.recorded = false;
    return 0;
}

NR_API float nr_profile_band_kernel_ms(void)
{
    BandKernelTimer &t = g_band_timer;
    float ms = -1.0f;
    if (!t.recorded || hipEventSynchronize(t.stop) != hipSuccess || hipEventElapsedTime(&ms, t.start, t.stop) != hipSuccess) {
        (void)hipGetLastError();
        return -1.0f;
    }
    return ms;
}
NR_API int nr_profile_band_kernel_which(void) { return g_band_timer.recorded ? g_band_timer.which : -1; }
// (pure host logic, no device needed: the per-launch rule as a function of the call's shape, for tests/test_abi.py)
NR_API int nr_profile_k6_choice(int32_t B, int32_t F, int32_t S, int32_t return_rgb, int32_t return_alpha, double eps, int32_t flags)
{
    size_t lds = 0, fl = 0;
    int win = 0, qcap = 0;
    const BandShape shape = band_shape(S);
    const bool fast_fits = fast_band_config(S, return_rgb != 0, shape, shape.w_max, &fl, &win, &qcap) != 0;
    (void)return_alpha;
    return k6_row_band(B, F, S, return_rgb != 0, eps, flags, fast_fits, &lds) > 0 ? 1 : 0;
}
#define NR_BAND_TIMER_START(st) if (g_band_timer.on) { g_band_timer.which = use_row ? 1 : 0; g_band_timer.recorded = hipEventRecord(g_band_timer.start, st) == hipSuccess; }
#define NR_BAND_TIMER_STOP(st) if (g_band_timer.on && g_band_timer.recorded) g_band_timer.recorded = hipEventRecord(g_band_timer.stop, st) == hipSuccess
#else
#define NR_BAND_TIMER_START(st) ((void)0)
#define NR_BAND_TIMER_STOP(st) ((void)0)
#endif

int nr::run_backward_pixel_map(const float *faces, const int32_t *face_index_map, const float *rgb_map,
                               const float *alpha_map, const float *grad_rgb_map, const float *grad_alpha_map,
                               float *grad_faces, int B, int F, int S, double eps, int return_rgb, int return_alpha,
                               int flags, const unsigned char *visible_faces, void *workspace, size_t workspace_bytes,
                               hipStream_t st, const int **vis_list_out, const int **vis_count_out,
                               const double **defer_scratch, const int **defer_slot_of, void *zero_ptr, size_t zero_bytes,
                               int *zeroed, const SetupHook *hook)
{
    if (zeroed) *zeroed = 0;
    if (vis_list_out) *vis_list_out = nullptr;
    if (vis_count_out) *vis_count_out = nullptr;
    if (defer_scratch) *defer_scratch = nullptr;
    if (defer_slot_of) *defer_slot_of = nullptr;
    if (!faces || !face_index_map || !grad_faces) return NR_E_NULL;
    if (!return_rgb && !return_alpha) return NR_E_MODE;  // rasterize.py:523-524 returns early; callers skip the call
    if (return_rgb && (!rgb_map || !grad_rgb_map)) return NR_E_NULL;
    if (return_alpha && (!alpha_map || !grad_alpha_map)) return NR_E_NULL;
    if (int e = check_sizes(B, F, S)) return e;
    if ((size_t)B * S * S > 0x7fffffffull / 3) return NR_E_SIZE;  // int32 pixel indexing inside the kernels
    const int n = B * F;
    const bool rgb = return_rgb != 0, alpha = return_alpha != 0;
    const bool exact = (flags & NR_FLAG_EXACT_GRADIENT) != 0;

    size_t lds = 0;
    int win = BAND_WIN, qcap = 0;
    // Narrower bands when the launch would have few band workgroups (small batches): the chip holds 768 of them at a time and
    // half of a teapot view's bands are empty; 16 views: stage 113 -> 104 us with W = 2, 4 views 70 -> 50, 1 view 66 -> 40 (W = 1).
    const BandShape shape = band_shape(S);
    int w_max = shape.w_max;
    while (w_max > 1 && (size_t)B * 2 * ((S + w_max - 1) / w_max) < 3072) w_max >>= 1;
    const int W_fast = fast_band_config(S, rgb, shape, w_max, &lds, &win, &qcap);
    size_t row_lds = 0;
    const int W_row = k6_row_band(B, F, S, rgb, eps, flags, W_fast != 0, &row_lds);  // (which launches take k_bpm_row: there)
    const bool use_row = W_row > 0;
    const int W = use_row ? 1 : W_fast;  // the band width of the tables
    if (W_fast == 0 || (flags & NR_FLAG_K6_GLOBAL)) {  // raster too large for an LDS band (or the fallback asked for: tests)
        const dim3 grid((unsigned)n), block(WAVE);
        if (rgb && alpha)
            hipLaunchKernelGGL((k_bpm_global<true, true>), grid, block, 0, st, faces, face_index_map, rgb_map,
                               alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, F, S, eps);
        else if (rgb)
            hipLaunchKernelGGL((k_bpm_global<true, false>), grid, block, 0, st, faces, face_index_map, rgb_map,
                               alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, F, S, eps);
        else
            hipLaunchKernelGGL((k_bpm_global<false, true>), grid, block, 0, st, faces, face_index_map, rgb_map,
                               alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, F, S, eps);
        return launch_status();
    }

    const BpmLayout L = bpm_layout(B, F, S);
    if (!workspace || workspace_bytes < L.total) return NR_E_WORKSPACE;
    const bool defer = defer_scratch && defer_slot_of;  // the caller's gather finishes the listed faces (see nr_device.h)
    unsigned char *ws = (unsigned char *)workspace;
    int *band_lines = (int *)(ws + L.band_off);
    const int n_bands = (S + W - 1) / W;
    double *scratch = (double *)(ws + L.scratch_off);
    int *vis_count = (int *)(ws + L.count_off);
    int *vis_list = (int *)(ws + L.list_off);
    int *slot_of = (int *)(ws + L.slot_off);
    unsigned *rng = (unsigned *)(ws + L.rng_off);
    if (vis_list_out) *vis_list_out = vis_list;
    if (vis_count_out) *vis_count_out = vis_count;
    const unsigned char *vflags = visible_faces;
    if (!vflags) {  // the forward's flags were not kept: one pass over face_index_map rebuilds them
        unsigned char *f = ws + L.flags_off;
        const int he = fill_bytes(f, 0, (size_t)n, st);
        if (he != 0) return he;
        const size_t P = (size_t)B * S * S;
        hipLaunchKernelGGL(k_mark_visible, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, face_index_map, f, F,
                           S * S, P);
        vflags = f;
    }
    int *band_start = (int *)(ws + L.start_off), *band_cursor = (int *)(ws + L.cursor_off), *lines_ok = (int *)(ws + L.ok_off);
    BandLine *line_buf = (BandLine *)(ws + L.lines_off);
    // line records from k_line_setup, unless NR_FLAG_K6_SCAN asks for the in-kernel face scan (tests) or the launch is
    // outside k_line_setup's shape (grid.y, 72 KB of LDS)
    const bool use_records = !(flags & NR_FLAG_K6_SCAN) && B <= 65535 && n_bands <= 3072;
    // the distance coefficients of a record: x 2 / S up front in the tolerance mode, as the reference has them in the exact one
    const float k2s = exact ? 1.0f : 2.0f / (float)S;
    const size_t cap = use_records ? L.cap : 0;  // capacity 0: every image is told to take the scan path
    int n_sum = 0;  // chunk rows per image that the consumer adds up (0: the band table is ready)
    // (k_compact_par keeps its chunk's 2 * n_bands line counters in LDS: 32 KB at most)
    if (L.n_chunks <= SMALL_CHUNKS && n_bands <= 4096) {
        int *chunk_band = (int *)(ws + L.cband_off);
        n_sum = L.n_chunks;
        // (line-difference counting while the two line arrays fit beside the band counters: raster sides up to 3583)
        const int use_diff = (size_t)(2 * n_bands + 2 * (S + 1)) * sizeof(int) <= 40960 ? 1 : 0;
        hipLaunchKernelGGL(k_compact_par, dim3((unsigned)L.n_chunks, (unsigned)B), dim3(VIS_CHUNK),
                           (size_t)(2 * n_bands + (use_diff ? 2 * (S + 1) : 0)) * sizeof(int), st, vflags, vis_list, vis_count,
                           slot_of, F, L.n_chunks, faces, rng, scratch, S, chunk_band, n_bands, W, band_cursor,
                           defer ? grad_faces : (float *)nullptr, hook ? 1 : 0, use_diff);
        if (!use_records)
            hipLaunchKernelGGL(k_band_total, dim3((unsigned)B), dim3(256), 0, st, chunk_band, n_sum, band_lines, band_start,
                               lines_ok, n_bands);
    } else {
        int *chunk_count = (int *)(ws + L.chunk_off);
        hipLaunchKernelGGL(k_count_visible, dim3((unsigned)L.n_chunks, (unsigned)B), dim3(VIS_CHUNK), 0, st, vflags,
                           chunk_count, F, L.n_chunks, band_lines, n_bands);
        const int lds_counters = n_bands <= 4096;  // 32 KB of LDS at most
        hipLaunchKernelGGL(k_compact_visible, dim3((unsigned)L.n_chunks, (unsigned)B), dim3(VIS_CHUNK),
                           lds_counters ? (size_t)2 * n_bands * sizeof(int) : 0, st, vflags, chunk_count, vis_list, vis_count,
                           slot_of, F, L.n_chunks, faces, rng, scratch, S, band_lines, n_bands, W, lds_counters,
                           defer ? grad_faces : (float *)nullptr, hook ? 1 : 0);
        hipLaunchKernelGGL(k_band_scan, dim3((unsigned)B), dim3(256), 0, st, band_lines, band_start, band_cursor, lines_ok,
                           n_bands, cap, 0);
    }
    {
        const LineSetupArgs la = {faces, face_index_map, vis_list, vis_count, rng, (const int *)(ws + L.cband_off), n_sum, band_lines,
                                  band_start, band_cursor, lines_ok, line_buf, L.cap, F, S, W, n_bands, k2s,
                                  (unsigned)((F + LS_FACES - 1) / LS_FACES), (unsigned)B, (size_t)6 * n_bands * sizeof(int)};
        if (hook) {  // the caller launches the line setup, together with its gather (nr_band_lines.h)
            if (int rc = hook->launch(hook->ctx, use_records ? &la : nullptr, vis_list, vis_count, slot_of, st)) return rc;
        } else if (use_records) {
            if (int rc = run_line_setup(la, st)) return rc;
        }
    }
    // lines per window: the packed piece scan keeps each class count in 16 bits (<= win * 2 * S / FSEG pieces)
    const int win_lines = min(win, max(4, min(BAND_WIN, (int)(65535ll * FSEG / (2ll * S))) & ~3));
    int rc;
    {
        // (a fill that rides in the band kernel: 16-byte words, and a slice per workgroup that is small next to the
        // workgroup's own work -- k6::FOLD_KB per band workgroup: 4 KB at the headline size, 61 KB on config 4; config 5's
        // 4 GB would be 2 MB for each of 2048 workgroups and go at 7 TB/s through a fill launch instead)
        const int W_band = use_row ? W_row : W_fast;  // lines per band workgroup of the kernel that carries the fill
        const size_t band_wgs = (size_t)((S + W_band - 1) / W_band) * 2 * (size_t)B;
        // (per workgroup: a 256-thread workgroup takes half of what a 512-thread one does)
        const bool zero_ok = !hook && zero_ptr && zero_bytes > 0 && zero_bytes % 16 == 0 && ((size_t)zero_ptr & 15) == 0 &&
                             zero_bytes <= band_wgs * ((size_t)k6::FOLD_KB << 10) * (size_t)(use_row ? rowk::NT : shape.threads) / 512;
        const int mode = !exact ? K6_FAST : ((S & (S - 1)) == 0 ? K6_EXACT_POW2 : K6_EXACT);
        auto launch = [&](auto r, auto a, auto m, auto nt) {
            constexpr bool R = decltype(r)::value, A = decltype(a)::value;
            constexpr int M = decltype(m)::value, NTH = decltype(nt)::value;
            // (behind k_bpm_row: only the images whose records exceed the line buffer, by the scan path, no fill)
            if (use_row)
                return launch_fast<R, A, M, NTH, true>(
                    faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, vis_list, vis_count, rng, scratch,
                    band_lines, band_start, lines_ok, line_buf, L.cap, B, F, S, W_fast, lds, eps, k2s, win_lines, qcap, st, nullptr, 0);
            return launch_fast<R, A, M, NTH>(
                faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, vis_list, vis_count, rng, scratch,
                band_lines, band_start, lines_ok, line_buf, L.cap, B, F, S, W_fast, lds, eps, k2s, win_lines, qcap, st,
                zero_ok ? zero_ptr : nullptr, zero_ok ? zero_bytes : 0);
        };
        using T = std::true_type;
        using N = std::false_type;
        auto by_threads = [&](auto r, auto a, auto m) {
            if (shape.threads == 256) return launch(r, a, m, std::integral_constant<int, 256>());
            return launch(r, a, m, std::integral_constant<int, 512>());
        };
        auto by_mode = [&](auto r, auto a) {
            if (mode == K6_FAST) return by_threads(r, a, std::integral_constant<int, K6_FAST>());
            if (mode == K6_EXACT_POW2) return by_threads(r, a, std::integral_constant<int, K6_EXACT_POW2>());
            return by_threads(r, a, std::integral_constant<int, K6_EXACT>());
        };
        NR_BAND_TIMER_START(st);
        rc = 0;
        if (use_row) {
            auto lpm = [&](auto r, auto a, auto m) {
                return launch_row<decltype(r)::value, decltype(a)::value, decltype(m)::value>(
                    face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, scratch, band_lines, band_start, lines_ok,
                    line_buf, L.cap, B, F, S, W_row, row_lds, eps, st, zero_ok ? zero_ptr : nullptr, zero_ok ? zero_bytes : 0);
            };
            auto lp = [&](auto r, auto a) {
                if (mode == K6_FAST) return lpm(r, a, std::integral_constant<int, K6_FAST>());
                if (mode == K6_EXACT_POW2) return lpm(r, a, std::integral_constant<int, K6_EXACT_POW2>());
                return lpm(r, a, std::integral_constant<int, K6_EXACT>());
            };
            rc = (rgb && alpha) ? lp(T(), T()) : (rgb ? lp(T(), N()) : lp(N(), T()));
            NR_BAND_TIMER_STOP(st);
        }
        if (rc == 0) rc = (rgb && alpha) ? by_mode(T(), T()) : (rgb ? by_mode(T(), N()) : by_mode(N(), T()));
        if (!use_row) NR_BAND_TIMER_STOP(st);
        if (rc == 0 && zero_ok && zeroed) *zeroed = 1;
    }
    if (rc) return rc;
    if (defer) {  // the caller finishes the listed faces (the fused gather, in the same launch as K7 / K8)
        *defer_scratch = scratch;
        *defer_slot_of = slot_of;
        return launch_status();
    }
    hipLaunchKernelGGL(k_bpm_finalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scratch, slot_of, grad_faces,
                       F, n, 0);
    return launch_status();
}

void nr::run_bpm_finalize(const double *scratch, const int *slot_of, float *grad_faces, int B, int F, hipStream_t st, bool add)
{
    const int n = B * F;
    hipLaunchKernelGGL(k_bpm_finalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scratch, slot_of, grad_faces,
                       F, n, add ? 1 : 0);
}

int nr::run_line_setup(const LineSetupArgs &a, hipStream_t st)
{
    static LdsLimit ls_limit;
    if (int rc = ls_limit.ensure((const void *)k_line_setup, a.lds_bytes)) return rc;
    hipLaunchKernelGGL(k_line_setup, dim3(a.grid_x, a.grid_y), dim3(256), a.lds_bytes, st, a);
    return launch_status();
}

NR_API int nr_backward_pixel_map(const float *faces, const int32_t *face_index_map, const float *rgb_map,
                                 const float *alpha_map, const float *grad_rgb_map, const float *grad_alpha_map,
                                 float *grad_faces, int32_t B, int32_t F, int32_t S, double eps, int32_t return_rgb,
                                 int32_t return_alpha, int32_t flags, const uint8_t *visible_faces, void *workspace,
                                 size_t workspace_bytes, void *stream)
{
    return run_backward_pixel_map(faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, grad_faces, B,
                                  F, S, eps, return_rgb, return_alpha, flags, visible_faces, workspace, workspace_bytes,
                                  (hipStream_t)stream, nullptr, nullptr);
}

// The fused backward's hook into K6 (nr_band_lines.h SetupHook): where K6 would launch its line setup, the K7 / K8 gather goes
// out with it in one grid (nr_backward_gather.hip, k_setup_gather).
namespace {
struct GatherCall {
    const float *faces, *faces_z_ref, *weight_map, *depth_map, *grad_rgb_map, *grad_depth_map;
    const int32_t *face_index_map;
    float *grad_textures, *grad_faces;
    int B, F, S, ts, flags;
    double eps;
    const FaceLight *lit;
    bool called;
    int depth_done;
};

int launch_setup_and_gather(void *ctx, const LineSetupArgs *ls, const int *vis_list, const int *vis_count, const int *slot_of,
                            hipStream_t st)
{
    GatherCall &g = *static_cast<GatherCall *>(ctx);
    g.called = true;
    // (no K6 scratch to finish: the gather adds K8's sums onto the zeros of the compaction, K6's follow behind the band kernel)
    return run_backward_textures(g.face_index_map, nullptr, nullptr, g.faces, g.faces_z_ref, g.weight_map, g.depth_map,
                                 g.grad_rgb_map, g.grad_textures, g.B, g.F, g.S, g.ts, g.eps, g.flags, vis_list, vis_count, st,
                                 g.grad_depth_map, g.grad_faces, &g.depth_done, nullptr, nullptr, nullptr, *g.lit, false, 1, ls,
                                 slot_of);
}
}  // namespace

// Fused backward: K6, K7 and K8 of one Rasterize.backward_gpu call (rasterize.py:849-889) behind one entry point.
// Same results as calling the three stage functions in the reference's order; the visible-face lists built for
// K6 are reused by the K7 / K8 gathers, which then visit ~1/5 of the faces.
NR_API int nr_backward_rasterize(const float *faces, const float *faces_z_ref, const int32_t *face_index_map,
                                 const float *weight_map, const float *depth_map, const float *rgb_map,
                                 const float *alpha_map, const float *grad_rgb_map, const float *grad_alpha_map,
                                 const float *grad_depth_map, float *grad_faces, float *grad_textures, int32_t B,
                                 int32_t F, int32_t S, int32_t ts, double eps, int32_t flags,
                                 const uint8_t *visible_faces, void *workspace, size_t workspace_bytes, void *stream)
{
    return nr_backward_rasterize_lit(nullptr, faces, faces_z_ref, face_index_map, weight_map, depth_map, rgb_map, alpha_map,
                                     grad_rgb_map, grad_alpha_map, grad_depth_map, grad_faces, grad_textures, B, F, S, ts,
                                     eps, flags, visible_faces, workspace, workspace_bytes, stream);
}

NR_API int nr_backward_rasterize_lit(const nr_face_light *lit, const float *faces, const float *faces_z_ref,
                                     const int32_t *face_index_map, const float *weight_map, const float *depth_map,
                                     const float *rgb_map, const float *alpha_map, const float *grad_rgb_map,
                                     const float *grad_alpha_map, const float *grad_depth_map, float *grad_faces,
                                     float *grad_textures, int32_t B, int32_t F, int32_t S, int32_t ts, double eps,
                                     int32_t flags, const uint8_t *visible_faces, void *workspace, size_t workspace_bytes,
                                     void *stream)
{
    if (!faces || !face_index_map || !grad_faces) return NR_E_NULL;
    if (int e = check_sizes(B, F, S)) return e;
    FaceLight fl;  // per-face light colours: only the texture gather sees them (the geometry gradients do not)
    if (int e = face_light_args(grad_rgb_map && grad_textures ? lit : nullptr, F, true, fl)) return e;
    hipStream_t st = (hipStream_t)stream;
    const bool use_rgb = grad_rgb_map != nullptr, use_alpha = grad_alpha_map != nullptr, use_depth = grad_depth_map != nullptr;
    const int *vis_list = nullptr, *vis_count = nullptr;
    // K6's last step (rounding the double sums into grad_faces, zeros for the unlisted faces) rides in the K7 / K8 gather's
    // launch when there is one that walks faces (texture_size <= 13)
    const bool fold = use_rgb && grad_textures && ts >= 2 && ts <= 13;
    const double *k6_scratch = nullptr;
    const int *k6_slot_of = nullptr;
    int tex_zeroed = 0;
    bool k6_done = false;
    // Small launches (up to 96 k faces in the call: 16 views of the 4928-face teapot) take the order
    //   compaction | line setup + gather + zeros of grad_textures in ONE grid | band kernel | the faces the gather left
    //   out + K6's sums onto grad_faces (one launch)
    // where the line setup and the gather -- two chains of dependent round trips that need nothing of each other -- run side
    // by side (8 views: backward 82 -> 72 us, 16: 111 -> 104; 32: 156 -> 152, not taken).  Larger ones keep
    //   compaction | line setup | band kernel with the fill on the side | gather with K6's finish:
    // there both launches are bound by how many workgroups the chip holds, a shared grid takes the sum of their times (64
    // views: 254.7 us either way), and the fill inside the band kernel and the finish inside the gather are worth more
    // (config 4: 0.80 vs 0.87 ms, 1024 views of 32 x 32: 0.72 vs 0.85, config 5 with its 4 GB of zeros: 1.57 vs 1.97).
    if (fold && (size_t)B * F <= k6::SHARED_LAUNCH_MAX_FACES && !(flags & NR_FLAG_SERIAL_BACKWARD)) {
        GatherCall gc = {faces, faces_z_ref, weight_map, depth_map, grad_rgb_map, use_depth ? grad_depth_map : nullptr,
                         face_index_map, grad_textures, grad_faces, B, F, S, ts, flags, eps, &fl, false, 0};
        const SetupHook hook = {&launch_setup_and_gather, &gc};
        if (int rc = run_backward_pixel_map(faces, face_index_map, rgb_map, use_alpha ? alpha_map : nullptr, grad_rgb_map,
                                            grad_alpha_map, grad_faces, B, F, S, eps, 1, use_alpha, flags, visible_faces, workspace,
                                            workspace_bytes, st, &vis_list, &vis_count, &k6_scratch, &k6_slot_of, nullptr, 0,
                                            nullptr, &hook))
            return rc;
        if (gc.called) {
            int dd = 0, finalized = 0;  // (K6's sums go onto grad_faces in k_backward_big's launch when there is one)
            if (int rc = run_backward_textures(face_index_map, nullptr, nullptr, faces, faces_z_ref, weight_map, depth_map,
                                               grad_rgb_map, grad_textures, B, F, S, ts, eps, flags, vis_list, vis_count, st,
                                               use_depth ? grad_depth_map : nullptr, grad_faces, &dd, k6_scratch, k6_slot_of,
                                               &finalized, fl, false, 2))
                return rc;
            if (use_depth && !gc.depth_done)
                if (int rc = run_backward_depth_map(faces, depth_map, face_index_map, nullptr, weight_map, grad_depth_map,
                                                    grad_faces, B, F, S, vis_list, vis_count, st, visible_faces))
                    return rc;
            if (k6_scratch && !finalized) run_bpm_finalize(k6_scratch, k6_slot_of, grad_faces, B, F, st, true);
            return launch_status();
        }
        // the band pipeline did not run (global-memory kernel: grad_faces complete, no lists): the gathers below, as they are
        k6_scratch = nullptr;
        k6_slot_of = nullptr;
        k6_done = true;
    }
    if (k6_done) {
    } else     if (use_rgb || use_alpha) {
        if (int rc = run_backward_pixel_map(faces, face_index_map, use_rgb ? rgb_map : nullptr,
                                            use_alpha ? alpha_map : nullptr, grad_rgb_map, grad_alpha_map, grad_faces,
                                            B, F, S, eps, use_rgb, use_alpha, flags, visible_faces, workspace,
                                            workspace_bytes, st, &vis_list, &vis_count, fold ? &k6_scratch : nullptr,
                                            fold ? &k6_slot_of : nullptr,
                                            // the zero fill of grad_textures inside the band kernel (with per-face light
                                            // colours: the cubes of the original faces)
                                            fold ? grad_textures : nullptr,
                                            (size_t)B * (fl.light ? fl.tex_faces : F) * ts * ts * ts * 3 * sizeof(float),
                                            &tex_zeroed))
            return rc;
    } else {
        const int e = fill_bytes(grad_faces, 0, (size_t)B * F * 9 * sizeof(float), st);  // :851
        if (e != 0) return e;
        // depth only: no K6 and therefore no lists -- built from the forward's flags when there are any (one launch), so that
        // the K8 gather visits the ~1/6 of the faces that own a pixel
        const ListsLayout L = lists_layout(B, F);
        const int n_chunks = (F + VIS_CHUNK - 1) / VIS_CHUNK;
        if (use_depth && visible_faces && workspace && workspace_bytes >= L.total && n_chunks <= SMALL_CHUNKS) {
            unsigned char *ws = (unsigned char *)workspace;
            int *list = (int *)(ws + L.list_off), *count = (int *)(ws + L.count_off);
            hipLaunchKernelGGL(k_list_visible, dim3((unsigned)n_chunks, (unsigned)B), dim3(VIS_CHUNK), 0, st, visible_faces,
                               list, count, F, n_chunks);
            vis_list = list;
            vis_count = count;
        }
    }
    int depth_done = 0;
    if (use_rgb && grad_textures) {
        // when both gradients are wanted, K8 rides along in the K7 gather (one walk of each face's screen box)
        int finalized = 0;
        if (int rc = run_backward_textures(face_index_map, nullptr, nullptr, faces, faces_z_ref, weight_map, depth_map,
                                           grad_rgb_map, grad_textures, B, F, S, ts, eps, flags, vis_list, vis_count, st,
                                           use_depth ? grad_depth_map : nullptr, grad_faces, &depth_done, k6_scratch,
                                           k6_slot_of, &finalized, fl, tex_zeroed != 0))
            return rc;
        if (k6_scratch && !finalized) run_bpm_finalize(k6_scratch, k6_slot_of, grad_faces, B, F, st);  // (not expected)
    }
    if (use_depth && !depth_done) {
        if (int rc = run_backward_depth_map(faces, depth_map, face_index_map, nullptr, weight_map, grad_depth_map,
                                            grad_faces, B, F, S, vis_list, vis_count, st, visible_faces))
            return rc;
    }
    return 0;
}

