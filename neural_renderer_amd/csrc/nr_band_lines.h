// nr_band_lines.h -- the line records of K6's band pipeline and the kernel body that writes them.
//
// Shared by nr_backward_pixel_map.hip (k_line_setup, the band kernel's scan path) and nr_backward_gather.hip, whose fused
// launch k_setup_gather runs the line setup and the K7 / K8 gather side by side: both need only the visible-face lists, both
// are latency-bound launches that leave most of the chip idle, and the band kernel between them needs the records alone.
#pragma once
#include "nr_device.h"

namespace nr {

struct __attribute__((aligned(16))) BandLine {
    int in_rng;   // from | to << 16 (from > to: empty)
    int out_rng;  // from | to << 16
    int geo;      // d1_in | ld << 16 | flags << 24   (flags: 1 has out, 2 has0, 4 has1, 8 direction > 0)
    int tgt;      // list position | v0 << 28 | v1 << 30
    float cross, c0, c1;
    int fn;
};

// (everything k_line_setup takes, as one block: the fused backward launches the body from a kernel of its own, beside the
// gather -- nr_backward_gather.hip, k_setup_gather)
struct LineSetupArgs {
    const float *faces;
    const int32_t *fi_map;
    const int *vis_list, *vis_count;
    const unsigned *rng;
    const int *chunk_band;
    int n_sum;
    int *band_lines, *band_start, *band_cursor, *lines_ok;
    BandLine *line_buf;
    size_t cap;
    int F, S, W, n_bands;
    float k2s;
    unsigned grid_x, grid_y;  // the launch: ceil(F / LS_FACES) x B workgroups of 256 threads, lds_bytes of dynamic LDS
    size_t lds_bytes;
};

// k_line_setup as a launch of its own (nr_backward_pixel_map.hip)
int run_line_setup(const LineSetupArgs &a, hipStream_t st);

// How the fused backward takes over the line-setup launch of run_backward_pixel_map: called where k_line_setup would be
// launched -- the visible-face lists exist in stream order -- with the launch's arguments (NULL when the band kernel derives
// its lines itself: NR_FLAG_K6_SCAN, shapes outside k_line_setup's).  With a hook the compaction zeroes grad_faces of EVERY
// face (the gather adds K8's sums before K6's arrive: run_bpm_finalize(add = true) afterwards) and nothing is filled on the
// side (zero_ptr is ignored).
struct SetupHook {
    int (*launch)(void *ctx, const LineSetupArgs *ls, const int *vis_list, const int *vis_count, const int *slot_of,
                  hipStream_t st);  // slot_of: face -> list position or -1, [B][F]
    void *ctx;
};

namespace {


// Line range of one edge along one axis (rasterize.py:567-569), packed lo | hi << 16; RNG_EMPTY (lo > hi) when the edge
// crosses no integer line or is parallel to the sweeps (p0x == p1x: both contributions are skipped, :648, :653).
constexpr unsigned RNG_EMPTY = 1u;

__device__ __forceinline__ unsigned edge_range(float p0x, float p1x, int S)
{
    const int d0_from = (int)fmax((double)ceilf(fminf(p0x, p1x)), 0.0);   // :568
    const int d0_to = (int)fmin((double)fmaxf(p0x, p1x), S - 1.0);        // :569
    return (p0x != p1x && d0_to >= d0_from) ? (unsigned)d0_from | ((unsigned)d0_to << 16) : RNG_EMPTY;
}


// --------------------------------------------------------------------------------------------------
// The line records of the band kernel.  Which pixels a sweep visits is decided here, with the reference's arithmetic, and
// must not depend on the arithmetic mode of the terms.
// One line record (rasterize.py:543-579, :604-609, :665-672; the reference's arithmetic: the crossing
// points decide WHICH pixels are visited, which must not depend on the mode).  fv: the face's 9 floats; (e, axis, d0): the
// line; ld = d0 - first line of its band; owner_of(d1): face index of pixel (d0, d1) along the axis.
// in two steps, so that a caller can have the ownership reads of several lines in flight before it finishes any of them
struct LineHead {
    float p0x, p0y, p1x, p1y, p2x, p2y, d0f, d1_cross;
    int direction, d1_in, d1_out;
    bool live;  // both the in and the out pixel lie inside the image (:578-579)
};

__device__ __forceinline__ LineHead fast_line_head(const float *__restrict__ fv, int e, int axis, int d0, int S)
{
    const float fs = (float)S;
    const int i0 = e, i1 = (e + 1) % 3, i2 = (e + 2) % 3;
    float fp[6];
#pragma unroll
    for (int k = 0; k < 3; k++) { fp[k] = to_pixel(fv[3 * k], fs); fp[3 + k] = to_pixel(fv[3 * k + 1], fs); }
    const int ox = axis ? 3 : 0, oy = axis ? 0 : 3;  // p[num][dim] = pp[num][(dim + axis) % 2] (:556)
    LineHead h;
    h.p0x = fp[ox + i0]; h.p0y = fp[oy + i0]; h.p1x = fp[ox + i1]; h.p1y = fp[oy + i1];
    h.p2x = fp[ox + i2]; h.p2y = fp[oy + i2];
    if (axis == 0) h.direction = (h.p0x < h.p1x) ? -1 : 1; else h.direction = (h.p0x < h.p1x) ? 1 : -1;  // :559-564
    h.d0f = (float)d0;
    h.d1_cross = (h.p1y - h.p0y) / (h.p1x - h.p0x) * (h.d0f - h.p0x) + h.p0y;                  // :573
    h.d1_in = (0 < h.direction) ? (int)floorf(h.d1_cross) : (int)ceilf(h.d1_cross);             // :574
    h.d1_out = h.d1_in + h.direction;                                                           // :575
    h.live = !(h.d1_in < 0 || S <= h.d1_in) && !(h.d1_out < 0 || S <= h.d1_out);                // :578-579
    return h;
}

// owner: face index of the line's in pixel (d0, d1_in) (only read when h.live); k2s: what the two distance coefficients are
// multiplied by up front -- 2 / S for the tolerance mode (:649 `* 2. / is` folded in), 1 for the exact one
__device__ __forceinline__ BandLine fast_line_finish(const LineHead &h, int ld, int S, int rfn, int tgt, int owner, float k2s)
{
    const float p0x = h.p0x, p0y = h.p0y, p1x = h.p1x, p1y = h.p1y, p2x = h.p2x, p2y = h.p2y, d0f = h.d0f;
    const int direction = h.direction, d1_in = h.d1_in, d1_out = h.d1_out;
    BandLine r;
    r.in_rng = 1; r.out_rng = 1; r.geo = 0; r.tgt = tgt;
    r.cross = r.c0 = r.c1 = 0.0f;
    r.fn = rfn;
    if (h.live) {
        int flags = (0 < direction) ? 8 : 0;
        if (p1x != d0f) flags |= 2;
        if (p0x != d0f) flags |= 4;
        r.c0 = (p1x - p0x) / (p1x - d0f) * k2s;  // :649 leading factor, invariant along the sweep (x 2 / S: see k2s)
        r.c1 = (p1x - p0x) / (d0f - p0x) * k2s;  // :654
        if (owner == rfn) {                      // :604-609
            const int lim = (0 < direction) ? S - 1 : 0;
            const int o_from = max(min(d1_out, lim), 0), o_to = min(max(d1_out, lim), S - 1);
            r.out_rng = o_from | (o_to << 16);
            flags |= 1;
        }
        float d0_cross2;                         // :665-672
        if ((d0f - p0x) * (d0f - p2x) < 0)
            d0_cross2 = (p2y - p0y) / (p2x - p0x) * (d0f - p0x) + p0y;
        else
            d0_cross2 = (p1y - p2y) / (p1x - p2x) * (d0f - p2x) + p2y;
        const int lim2 = (0 < direction) ? (int)ceilf(d0_cross2) : (int)floorf(d0_cross2);
        const int i_from = max(min(d1_in, lim2), 0), i_to = min(max(d1_in, lim2), S - 1);
        r.in_rng = i_from | (i_to << 16);
        r.geo = d1_in | (ld << 16) | (flags << 24);
        r.cross = h.d1_cross;
    }
    return r;
}

template <typename OwnerOf>
__device__ __forceinline__ BandLine make_fast_line(const float *__restrict__ fv, int e, int axis, int d0, int ld, int S,
                                                   int rfn, int tgt, OwnerOf owner_of, float k2s)
{
    const LineHead h = fast_line_head(fv, e, axis, d0, S);
    return fast_line_finish(h, ld, S, rfn, tgt, h.live ? owner_of(h.d1_in) : -1, k2s);
}

// --------------------------------------------------------------------------------------------------
// k_line_setup: the line records of every (visible face, edge, axis, line d0), written band by band into line_buf so that
// a band workgroup finds its lines as one dense array: no face scan, no record compaction, no line setup inside the band
// kernel (together ~40 % of its cycles when they ran there, on <= 256 of its 512 threads).
//   One workgroup takes LS_FACES list positions of one image (dealt out in turn, see the kernel).  Binning without a
//   device-wide atomic per line (1.2 M same-address atomics across the 8 L2s of the chip cost 230 us): (1) the workgroup
//   counts its own lines per band in LDS, (2) reserves its block of each non-empty band with ONE global atomic (band_cursor),
//   (3) computes the records and places each at band start + block base + an LDS cursor.  The order inside a band is
//   irrelevant: every record is accumulated independently.  tgt = list position | v0 << 28 | v1 << 30.
//   The band table (lines per band, where each band starts) is the sum of the rows k_compact_par left per chunk: every
//   workgroup adds them up for itself, the image's first one also publishes the table for the band kernel.
// Images whose lines exceed the buffer's capacity (lines_ok[b] == 0) are skipped here and take the scan path of k_bpm_fast
// (every backward test forces that path as well: tests/test_hip_parity.py check_backward, NR_FLAG_K6_SCAN).
constexpr int LS_UNROLL = 4;  // lines per thread and round of k_line_setup
constexpr int LS_FACES = 32;  // list positions per workgroup (measured with a thread per item: 64 -> 43 us, 32 -> 29 us, 16 -> 29 us)

// Adds up the n_sum rows chunk_band[b][.][i] of an image (k_compact_par) into tot[i], i < n2, and takes the exclusive prefix
// start[i]; returns the image's total.  All 256 threads of the workgroup; tot / start are LDS arrays.
__device__ __forceinline__ int band_sum_prefix(const int *__restrict__ rows, int n_sum, int n2, int *tot, int *start,
                                               int *s_tmp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n2; i += 256) {
        int t = 0;
        for (int c = 0; c < n_sum; ++c) t += rows[(size_t)c * n2 + i];
        tot[i] = t;
    }
    __syncthreads();
    // thread t owns the entries [t * per, (t + 1) * per)
    const int per = (n2 + 255) / 256, i0 = tid * per, i1 = min(n2, i0 + per);
    int local = 0;
    for (int i = i0; i < i1; ++i) local += tot[i];
    int inc = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, WAVE);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    int run = inc - local;
    for (int w = 0; w < wave; ++w) run += s_tmp[w];
    for (int i = i0; i < i1; ++i) {
        start[i] = run;
        run += tot[i];
    }
    const int total = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
    __syncthreads();
    return total;
}

// bx, by: the workgroup's place in the line-setup grid (blockIdx of k_line_setup)
__device__ __forceinline__ void line_setup_body(const LineSetupArgs &a, const int bx, const int by)
{
    const float *__restrict__ faces = a.faces;
    const int32_t *__restrict__ fi_map = a.fi_map;
    const int *__restrict__ vis_list = a.vis_list, *__restrict__ vis_count = a.vis_count;
    const unsigned *__restrict__ rng = a.rng;
    const int *__restrict__ chunk_band = a.chunk_band;
    const int n_sum = a.n_sum;
    int *__restrict__ band_lines = a.band_lines, *__restrict__ band_start = a.band_start, *__restrict__ band_cursor = a.band_cursor,
        *__restrict__ lines_ok = a.lines_ok;
    BandLine *__restrict__ line_buf = a.line_buf;
    const size_t cap = a.cap;
    const int F = a.F, S = a.S, W = a.W, n_bands = a.n_bands;
    const float k2s = a.k2s;
    extern __shared__ int s_cnt[];  // [2 * n_bands] this workgroup's lines per band, then its fill cursors; [2 * n_bands] bases;
    int *s_base = s_cnt + 2 * n_bands;  // [2 * n_bands] where the image's bands start in its buffer
    int *s_start = s_base + 2 * n_bands;
    __shared__ int s_tmp[4];
    __shared__ unsigned s_rng[6 * LS_FACES];  // [position][axis * 3 + edge]
    __shared__ int s_lp[6 * LS_FACES + 1];    // first line of each item in the workgroup's numbering; total behind
    __shared__ int s_fn[LS_FACES];
    __shared__ float s_face[9 * LS_FACES];
    const int b = by, tid = threadIdx.x;
    const bool first = bx == 0;  // publishes the image's band table for the band kernel
    // The ceil(n_vis / LS_FACES) workgroups that have work deal the list positions out in turn (position = workgroup + k *
    // workgroups): neighbours in the list are neighbours in the mesh, and a block of 32 large faces has three times the lines
    // of an average one -- the kernel is one round of workgroups and as slow as its slowest.
    const int n_vis = vis_count[b];
    const int n_wg = (n_vis + LS_FACES - 1) / LS_FACES;
    const int pos0 = bx, pstep = n_wg;  // position of slot p: pos0 + p * pstep
    const int n_pos = bx < n_wg ? (n_vis - pos0 + pstep - 1) / pstep : 0;
    if (n_pos == 0 && !first) return;
    unsigned r_own = RNG_EMPTY;
    int fn_own = 0;
    if (tid < 6 * n_pos) {
        const int p = tid / 6, ae = tid - 6 * p, axis = ae / 3, e = ae - 3 * axis;
        r_own = rng[(((size_t)b * 2 + axis) * F + pos0 + p * pstep) * 3 + e];
    }
    if (tid < n_pos) fn_own = vis_list[(size_t)b * F + pos0 + tid * pstep];
    int ok;
    if (n_sum > 0) {
        // the image's lines per band = the sum of its chunk rows; every workgroup derives the band starts itself
        const int total = band_sum_prefix(chunk_band + (size_t)b * n_sum * 2 * n_bands, n_sum, 2 * n_bands, s_cnt, s_start, s_tmp);
        ok = (size_t)total <= cap ? 1 : 0;
        if (first) {
            for (int i = tid; i < 2 * n_bands; i += 256) {
                band_lines[(size_t)b * 2 * n_bands + i] = s_cnt[i];
                band_start[(size_t)b * 2 * n_bands + i] = s_start[i];
            }
            if (tid == 0) lines_ok[b] = ok;
        }
    } else {  // k_band_scan has prepared the table (large meshes)
        ok = lines_ok[b];
        for (int i = tid; i < 2 * n_bands; i += 256) s_start[i] = band_start[(size_t)b * 2 * n_bands + i];
    }
    if (n_pos == 0 || !ok) return;
    __syncthreads();
    for (int i = tid; i < 2 * n_bands; i += 256) s_cnt[i] = 0;
    if (tid < 6 * LS_FACES) s_rng[tid] = r_own;
    if (tid < LS_FACES) s_fn[tid] = fn_own;
    __syncthreads();
    // the vertices of the workgroup's faces (requested here, consumed after the reservations)
    float fv_a = 0.0f, fv_b = 0.0f;
    {
        const int p = tid / 9, k = tid - 9 * p;  // 256 threads: faces 0 .. 27 and 4 floats of face 28
        if (p < n_pos) fv_a = faces[((size_t)b * F + s_fn[p]) * 9 + k];
        const int t2 = tid + 256, p2 = t2 / 9, k2 = t2 - 9 * p2;
        if (t2 < 9 * LS_FACES && p2 < n_pos) fv_b = faces[((size_t)b * F + s_fn[p2]) * 9 + k2];
    }
    // (1) lines per band of this workgroup's (face, axis, edge) items; first line number of each item
    {
        int lines = 0;
        if (tid < 6 * n_pos) {
            const int p = tid / 6, ae = tid - 6 * p, axis = ae / 3;
            const unsigned pr = s_rng[tid];
            const int lo = (int)(pr & 0xffffu), hi = (int)(pr >> 16);
            for (int band = lo / W; band * W <= hi; ++band)  // lo > hi (RNG_EMPTY): no iteration
                atomicAdd(s_cnt + axis * n_bands + band, min(hi, band * W + W - 1) - max(lo, band * W) + 1);
            lines = hi >= lo ? hi - lo + 1 : 0;
        }
        const int lane = tid & 63, wave = tid >> 6;
        int inc = lines;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, WAVE);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_tmp[wave] = inc;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_tmp[w];
        if (tid < 6 * LS_FACES) s_lp[tid] = before + inc - lines;
        if (tid == 6 * LS_FACES - 1) s_lp[6 * LS_FACES] = before + inc;
    }
    __syncthreads();
    // (2) one reservation per non-empty band
    for (int i = tid; i < 2 * n_bands; i += 256) {
        const int c = s_cnt[i];
        s_base[i] = c > 0 ? s_start[i] + atomicAdd(band_cursor + (size_t)b * 2 * n_bands + i, c) : 0;
        s_cnt[i] = 0;
    }
    s_face[tid] = fv_a;
    if (tid + 256 < 9 * LS_FACES) s_face[tid + 256] = fv_b;
    __syncthreads();
    // (3) the records.  The workgroup's lines are numbered through (s_lp: first line of each item) and dealt to the threads
    // LS_UNROLL at a time: a thread first requests the ownership words of all its lines of the round, then finishes them (with
    // a thread per item walking its lines, every line waited for its own read: ~10 round trips in the longest item).
    const size_t img = (size_t)b * S * S;
    BandLine *buf_b = line_buf + (size_t)b * cap;
    const int n_items = 6 * n_pos, n_lines = s_lp[n_items];
    for (int base = 0; base < n_lines; base += LS_UNROLL * 256) {
        LineHead h[LS_UNROLL];
        int item[LS_UNROLL], d0v[LS_UNROLL], own[LS_UNROLL];
#pragma unroll
        for (int u = 0; u < LS_UNROLL; u++) {
            const int l = base + u * 256 + tid;
            item[u] = -1;
            own[u] = -1;
            if (l < n_lines) {
                int lo = 0, hi = n_items;  // last item with s_lp[item] <= l
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_lp[mid] <= l) lo = mid; else hi = mid;
                }
                item[u] = lo;
                const int p = lo / 6, ae = lo - 6 * p, axis = ae / 3, e = ae - 3 * axis;
                const int d0 = (int)(s_rng[lo] & 0xffffu) + (l - s_lp[lo]);
                d0v[u] = d0;
                float fv[9];
#pragma unroll
                for (int k = 0; k < 9; k++) fv[k] = s_face[9 * p + k];
                h[u] = fast_line_head(fv, e, axis, d0, S);
                if (h[u].live) own[u] = fi_map[axis ? img + (size_t)d0 * S + h[u].d1_in : img + (size_t)h[u].d1_in * S + d0];
            }
        }
#pragma unroll
        for (int u = 0; u < LS_UNROLL; u++) {
            if (item[u] < 0) continue;
            const int p = item[u] / 6, ae = item[u] - 6 * p, axis = ae / 3, e = ae - 3 * axis;
            const int band = d0v[u] / W, ld = d0v[u] - band * W;
            const int tgt = (pos0 + p * pstep) | (e << 28) | (((e + 1) % 3) << 30);
            const int bi = axis * n_bands + band;
            buf_b[s_base[bi] + atomicAdd(s_cnt + bi, 1)] = fast_line_finish(h[u], ld, S, s_fn[p], tgt, own[u], k2s);
        }
    }
}

}  // namespace
}  // namespace nr
