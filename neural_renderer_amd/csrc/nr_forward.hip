// nr_forward.hip -- forward kernels of the gfx950 rasterizer + their C-ABI entry points.
//
//   k_face_raster   F1+F2  per face: back-face cull, inverse matrix, screen box (ref K1, :240-277) and
//                          rasterization of the face's pixels into a packed 64-bit z-buffer (ref K2, :279-359)
//   k_large_raster  F2'    faces with a large screen box: one wave or one workgroup per face
//   k_resolve       F2''   per pixel: decode the winner, write face_index / weight / depth / face_inv maps (+ fused F3);
//   k_resolve_quads        the same with 16-byte fills of undrawn stretches (kept workspace, even raster side)
//   k_shade         F3  per pixel: trilinear texture sampling + background + alpha        (ref K4+K5, :361-465)
#include "nr_device.h"

using namespace nr;

namespace {

// --------------------------------------------------------------------------------------------------
// F2: face-parallel rasterizer with a packed 64-bit z-buffer.
//
// Measured history (profiles/r01a, r01b): a per-tile pixel-parallel kernel spent 0.8 ms on the headline scene
// although the arithmetic is ~10 us worth of VALU work, because the teapot concentrates hundreds of tiny
// faces in a few tiles (185 candidate faces over one 16x4 pixel block) and every candidate costs a
// wave-wide test; a tile-local face-parallel variant still left 4/5 of the chip idle (only ~18 of 64
// tiles per view contain geometry).  Faces, not pixels, are the balanced unit of work of a fine mesh:
//   k_face_raster    a wave takes 64 faces: back-face cull, inverse matrix, screen box (K1) one face per lane; the faces with
//                    a small box (<= SMALL_AREA pixels, i.e. practically every face of a mesh) are then rasterized by the
//                    wave together -- their rows, then their inside pixels are dealt to the lanes (see the kernel) -- each
//                    inside pixel evaluating the reference's barycentric / depth test (K2 body) and publishing
//                    (depth bits << 32 | face index) with a 64-bit atomicMin on the pixel's z-buffer word; faces with a
//                    larger box are queued;
//   k_large_raster   queued faces: a wave each up to WAVE_AREA pixels, a whole workgroup beyond (and for needle strips);
//   k_resolve        one thread per pixel decodes the winner, re-evaluates its inverse matrix and weights (same
//                    functions, same inputs -> same bits: the per-face inverses are never stored, which saves
//                    36 B per face written + 36 B per covered pixel re-read) and writes face_index / weight / depth /
//                    face_inv maps (every element, init values where no face was found, rasterize.py:478-496) and,
//                    on request, the per-face "owns a pixel" flags that the backward's K6 pipeline starts from.
// The packed minimum reproduces the reference's winner rule "smaller zp, ties -> lower face index" (it
// scans faces in ascending order with a strict `<`, rasterize.py:300,334).  The depth goes in as the usual
// order-preserving integer key of a float (depth_key: sign bit set for positive values, all bits flipped for
// negative ones), so any `near` -- the reference accepts any, :331 -- orders correctly, faces behind the camera
// (negative zp with near < 0) included.  The result does not depend on the order of the atomics:
// face_index_map is bit-reproducible.
#ifndef NR_FWD_SMALL_FACES  // (development knob)
#define NR_FWD_SMALL_FACES (16384 * 32)
#endif
constexpr size_t SMALL_LAUNCH_FACES = (size_t)NR_FWD_SMALL_FACES;  // launches of fewer faces take 16 faces per raster wave (k_face_raster)
constexpr int SMALL_AREA = 256;   // boxes up to this many pixels: rasterized by k_face_raster (measured with the round-2 form of
                                  // the kernel: 128 / 64 make config 4 15 % / 28 % slower, the headline +0 / +11 %)
constexpr int WAVE_AREA = 4096;   // up to this: one wave per face (wave_raster); beyond, and strips: one workgroup (k_large_raster)
constexpr unsigned long long ZEMPTY = ~0ull;
// The two queues of k_face_raster (faces for a wave each, faces for a workgroup each) are SHARDED: a raster workgroup appends to
// shard blockIdx & 15, every shard with its own pair of counters on its own 128 bytes.  One pair for the whole launch was the
// limit of dense meshes: a word takes ~88 returning atomics per microsecond (MI355X_MICROARCH.md, "dequeue"), and config 4 --
// 10 240 raster waves, nearly every one with a face to queue -- spent 66 of its 152 us there (profiles/r06_pmc_fwd_C4.txt:
// 54 % of the wave-cycles parked; with the queues switched off: 86 us).  16 shards: ~1400 atomics per microsecond, and a
// consumer workgroup reads 16 cache lines to learn that there is nothing to do (64 shards, every consumer WAVE reading 64
// lines: the headline's forward 72 -> 85 us for its empty queues).
constexpr int QSHARDS = 16;
constexpr int QSTRIDE = 32;  // ints between the counter pairs of neighbouring shards

// monotone float -> uint32 key: a < b  <=>  depth_key(a) < depth_key(b) for all non-NaN floats; -0 is keyed as +0 (the
// reference's `zp < depth_min` does not tell them apart, so the lower face index must win between them).  A candidate
// always satisfies zp < (float)far, so its key is below 0xff800000 and the packed word below ZEMPTY.
__device__ __forceinline__ unsigned depth_key(float zp)
{
    const unsigned u = __float_as_uint(zp + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// z-buffer word.  Classic (epoch < 0): depth key << 32 | face index, empty = all ones after the per-call fill.  Epoch mode
// (NR_FLAG_ZBUF_EPOCH, a workspace the caller keeps between calls): epoch << 56 | depth key << 24 | face index (< 2^24); the
// caller counts the epoch DOWN from call to call, so every word of an earlier call is larger than any word of this one and
// loses the atomic minimum -- no fill and no clean-up pass (3 x 33.5 MB of z-buffer traffic per forward become 2 x) -- and
// the resolve pass treats a word whose epoch is not the current one as empty.
__device__ __forceinline__ unsigned long long zword(float zp, unsigned fn, int epoch)
{
    const unsigned long long k = depth_key(zp);
    return epoch < 0 ? (k << 32) | fn : ((unsigned long long)(unsigned)epoch << 56) | (k << 24) | fn;
}

struct FaceGeo {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2, i0, i1, i2, i3, i4, i5, i6, i7, i8;
};

// The reference's K2 body in two parts, so that a raster loop can run the cheap half-plane tests over the pixels of a box and
// spend the expensive part (seven IEEE divisions) only on pixels inside the triangle -- with all lanes that found one.
// rasterize.py:310-312 (back faces never reach here: their box is empty)
__device__ __forceinline__ bool inside_edges(const FaceGeo &q, float xp, float yp)
{
    return !(((yp - q.y0) * (q.x1 - q.x0) < (xp - q.x0) * (q.y1 - q.y0)) ||
             ((yp - q.y1) * (q.x2 - q.x1) < (xp - q.x1) * (q.y2 - q.y1)) ||
             ((yp - q.y2) * (q.x0 - q.x2) < (xp - q.x2) * (q.y0 - q.y2)));
}

// weights and depth of a pixel inside the triangle; false when the depth test rejects it
__device__ __forceinline__ bool eval_inside(const FaceGeo &q, float xif, float yif, double near_d, double far_d, float &zp,
                                            float &w0, float &w1, float &w2)
{
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
    // :334 `zp < depth_min` with depth_min starting at (float)far: implied by :331 for ordinary numbers, but a NaN zp (NaN /
    // Inf vertices, zero depth) passes :331 and must never win a pixel
    return zp < (float)far_d;
}

// One (face, pixel) evaluation.  Returns false when the pixel is rejected.
__device__ __forceinline__ bool eval_pixel(const FaceGeo &q, float xp, float yp, float xif, float yif, double near_d,
                                           double far_d, float &zp, float &w0, float &w1, float &w2)
{
    return inside_edges(q, xp, yp) && eval_inside(q, xif, yif, near_d, far_d, zp, w0, w1, w2);
}

// pixel centre (2. * i + 1 - is) / is (rasterize.py:291-292, evaluated in double there).  Both operands are
// integers below 2^24, exactly representable in float, and a correctly rounded float division of exact
// operands equals the double division rounded to float (53 >= 2 * 24 + 2: the double rounding is innocuous).
__device__ __forceinline__ float pixel_center_f(int i, int S) { return (float)(2 * i + 1 - S) / (float)S; }
// same value with one multiply when S is a power of two (the quotient is then exactly representable)
__device__ __forceinline__ float pixel_center_p(int i, int S, float inv_s, bool pow2)
{
    return pow2 ? (float)(2 * i + 1 - S) * inv_s : (float)(2 * i + 1 - S) / (float)S;
}

// vertices -> FaceGeo with the inverse barycentric matrix of K1 (rasterize.py:240-277); zeros for back faces (:240, :253)
__device__ __forceinline__ void load_face_geo(const float *__restrict__ f, int S, FaceGeo &g, float inv[9])
{
    g.x0 = f[0]; g.y0 = f[1]; g.z0 = f[2]; g.x1 = f[3]; g.y1 = f[4]; g.z1 = f[5]; g.x2 = f[6]; g.y2 = f[7]; g.z2 = f[8];
    if (is_backside(g.x0, g.y0, g.x1, g.y1, g.x2, g.y2)) {
#pragma unroll
        for (int k = 0; k < 9; k++) inv[k] = 0.0f;
    } else {
        const float fs = (float)S;
        const float px[3] = {to_pixel(g.x0, fs), to_pixel(g.x1, fs), to_pixel(g.x2, fs)};
        const float py[3] = {to_pixel(g.y0, fs), to_pixel(g.y1, fs), to_pixel(g.y2, fs)};
        compute_face_inv(px, py, inv);
    }
    g.i0 = inv[0]; g.i1 = inv[1]; g.i2 = inv[2]; g.i3 = inv[3]; g.i4 = inv[4]; g.i5 = inv[5];
    g.i6 = inv[6]; g.i7 = inv[7]; g.i8 = inv[8];
}

// --------------------------------------------------------------------------------------------------
// k_face_raster: one lane per face for the per-face work, then the *wave* re-distributes the work of its 64 faces twice so that
// the expensive part runs with full lanes (a thread-per-face or 4-lanes-per-face loop over the box pays the seven IEEE divisions
// of the K2 body max-over-lanes of the box sizes with ~30 % of the lanes active: half of the faces are culled, boxes differ, and
// half of a box lies outside its triangle):
//   1. lane = face: cull, screen box, inverse matrix; faces with a small box ("kept") put their constants into the wave's LDS;
//   2. lane = (kept face, row of its box): the inside pixels of a row are one interval (each half-plane test is monotonic in xp
//      for a fixed yp, rounding included) -- found with the reference's tests alone (:310-312), a few cheap steps per lane;
//   3. lane = inside pixel: weights, depth, depth test, 64-bit atomicMin.
// Rows and pixels are handed over through small LDS lists, written by their producers at wave-prefix positions, a window of
// the list at a time (any box shape fits).  The waves of a workgroup are independent: no barrier, only wave-scope fences.
constexpr int FR_ROWS = 128;  // row items per window
constexpr int FR_PIX = 256;   // pixel items per window (7.4 KB of LDS per wave: five workgroups per CU)
// FACES: faces per wave (its first FACES lanes take one each) and GROUP: consecutive faces per run of the face -> lane mapping
// are template parameters: 64 / 16 for large launches, 16 / 4 below 16 384 x 32 faces (round 4: the kernel is latency-bound,
// a wave lives as long as its rows and pixels take, and smaller waves' worth of faces means more, shorter waves -- fused
// forward in us, teapot views at 256^2: 8 views 43 -> 32, 16 views 42 -> 34, 32 views 62 -> 47, 48 views 67 -> 58, 64 views
// 70 -> 70; 64 views at 512^2 306 -> 290; rounds 2-3 had 32 / 8 below 2048 x 64 faces).  Above the threshold 16 per wave loses:
// config 4 (655 360 faces) 242 -> 323, 256 views of 128^2 101 -> 132, 1024 views of 32^2 142 -> 265.
// Measured and dropped: two pixels per lane and evaluation step (neutral, more code); the row interval estimated from the edge
// equations and pinned down with ~4 exact tests by a one-test-per-step state machine (89 vs 81 us fused forward).
static_assert(SMALL_AREA <= 256, "rows and columns of a kept box are packed into 8 bits each");

template <int FACES>
struct FaceWaveLds {
    float g[18][FACES];  // x0 y0 x1 y1 x2 y2 | z0 z1 z2 | inv[9], component-major: lanes with different faces hit different banks
    int x_lo[FACES], y_lo[FACES], bw[FACES], img[FACES], fn[FACES];
    int rows[FR_ROWS];
    int pix[FR_PIX];
};

// inclusive prefix sum over the 64 lanes with DPP moves only (no LDS crossbar round trips): Hillis-Steele inside each row of 16
// lanes, then the row totals are passed on (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ int wave_inclusive_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return v;
}

// LDS hand-over between the lanes of one wave: the wave's LDS operations execute in order; the fence keeps the compiler from
// moving accesses across and waits for the writes
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// pixel item (slot << 16 | row << 8 | column) -> z-buffer word and key; false when the depth test rejects the pixel
template <int FACES>
__device__ __forceinline__ bool slot_pixel(const FaceWaveLds<FACES> &L, int e, int S, double near_d, double far_d,
                                           unsigned long long *__restrict__ zbuf, unsigned long long &key,
                                           unsigned long long *&at, int epoch)
{
    const int s = e >> 16;
    FaceGeo q;
    q.z0 = L.g[6][s]; q.z1 = L.g[7][s]; q.z2 = L.g[8][s];
    q.i0 = L.g[9][s]; q.i1 = L.g[10][s]; q.i2 = L.g[11][s]; q.i3 = L.g[12][s]; q.i4 = L.g[13][s];
    q.i5 = L.g[14][s]; q.i6 = L.g[15][s]; q.i7 = L.g[16][s]; q.i8 = L.g[17][s];
    const int x = L.x_lo[s] + (e & 255), y = L.y_lo[s] + ((e >> 8) & 255);
    float zp, w0, w1, w2;
    const bool ok = eval_inside(q, (float)x, (float)y, near_d, far_d, zp, w0, w1, w2);
    key = zword(zp, (unsigned)L.fn[s], epoch);
    at = zbuf + ((size_t)L.img[s] * S + y) * S + x;
    return ok;
}


template <bool POW2, int FACES, int GROUP>
__global__ __launch_bounds__(256) void k_face_raster(const float *__restrict__ faces,
                                                     unsigned long long *__restrict__ zbuf,
                                                     int *__restrict__ large_list, int *__restrict__ wave_list,
                                                     int *__restrict__ n_large, int qcap, int nshards,
                                                     unsigned char *__restrict__ visible_faces, int n_faces_total, int F,
                                                     int S, double near_d, double far_d, int epoch,
                                                     unsigned char *__restrict__ touched)
{
    __shared__ FaceWaveLds<FACES> lds[4];
    FaceWaveLds<FACES> &L = lds[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_faces_total && visible_faces) visible_faces[t] = 0;  // k_resolve raises the flags of the faces that win a pixel
    // A wave takes 64 / GROUP groups of GROUP consecutive faces, the groups W apart (W = number of waves): faces that are
    // neighbours in the mesh are neighbours on screen and of similar size, and a wave of 64 consecutive large ones has ten
    // times the average work -- the kernel then waits for a few waves (teapot view: 2566 inside pixels in the heaviest wave
    // against a mean of 233; fully interleaved: 280).
    const int n_waves = (n_faces_total + FACES - 1) / FACES;
    const int i = ((lane / GROUP) * n_waves + (t >> 6)) * GROUP + (lane % GROUP);
    const bool live = lane < FACES && (t >> 6) < n_waves && i < n_faces_total;
    // the wave's faces: FACES / GROUP runs of GROUP * 9 floats, fetched with consecutive lanes on consecutive words (a
    // lane fetching its own 36 bytes makes 64 requests per load instruction) and handed to their lanes through LDS
    float f[9];
    {
        float *stage = &L.g[0][0];  // half of its floats; the slot constants move in after the hand-over
        const size_t n_words = (size_t)n_faces_total * 9;
        // (all loads first, from clamped addresses, then the stores: a load under its own condition is a basic block of its own,
        // and each then waits for its data before the next is issued -- nine dependent round trips per wave at FACES = 64:
        // profiles/r06_pmc_fwd_C4.txt, 54 % of config 4's wave-cycles parked)
        constexpr int NLOAD = (9 * FACES + 63) / 64;
        float word[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; k++) {
            const int d = k * 64 + lane, run = d / (9 * GROUP), off = d - run * (9 * GROUP);
            const size_t src = ((size_t)run * n_waves + (t >> 6)) * (9 * GROUP) + off;
            const float v = faces[src < n_words ? src : n_words - 1];
            word[k] = src < n_words ? v : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NLOAD; k++) {
            const int d = k * 64 + lane;
            if (d < 9 * FACES) stage[d] = word[k];
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = stage[(lane & (FACES - 1)) * 9 + k];
        wave_lds_sync();
    }
    Cand cd = face_candidates(f[0], f[1], f[3], f[4], f[6], f[7], S);
    if (!live) cd.n = 0;
    // Boxes too large for this kernel are queued: medium ones for a wave each, strips (needles) and large ones for a whole
    // workgroup each (k_large_raster).  One atomic per wave and queue (a same-address atomic per face from all over the chip
    // serialises at the memory side), on the counters of the workgroup's shard (QSHARDS); all counters start at -1 (one fill
    // with the z-buffer).  A shard holds qcap entries: what its workgroups can queue at most.
    const bool queued = cd.n > 0 && (cd.strip || cd.n > SMALL_AREA);
    const bool to_wave = queued && !cd.strip && cd.n <= WAVE_AREA;
    const bool to_large = queued && !to_wave;
    {
#ifdef NR_FWD_NOQUEUE  // (development: queued faces are dropped -- what do the queues' counters cost?)
        const unsigned long long mw = 0, ml = 0;
#else
        const unsigned long long mw = __ballot(to_wave), ml = __ballot(to_large);
#endif
        const int shard = (int)(blockIdx.x & (unsigned)(nshards - 1));  // (1 or QSHARDS)
        int *qc = n_large + shard * QSTRIDE;
        if (mw) {
            const int leader = __ffsll((long long)mw) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(qc + 1, __popcll(mw)) + 1;
            base = __shfl(base, leader, WAVE);
            if (to_wave) wave_list[(size_t)shard * qcap + base + __popcll(mw & ((1ull << lane) - 1ull))] = i;
        }
        if (ml) {
            const int leader = __ffsll((long long)ml) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(qc, __popcll(ml)) + 1;
            base = __shfl(base, leader, WAVE);
            if (to_large) large_list[(size_t)shard * qcap + base + __popcll(ml & ((1ull << lane) - 1ull))] = i;
        }
    }
    // 1. kept faces -> slots (lane order), their constants -> LDS
    const bool keep = cd.n > 0 && !queued;  // not: back faces, off-screen faces, coincident vertices; queued faces
    const unsigned long long mk = __ballot(keep);
    if (mk == 0) {
        return;
    }
    const int slot = __popcll(mk & ((1ull << lane) - 1ull));
    const int bh = keep ? cd.n / cd.bw : 0;
    if (keep) {
        FaceGeo g;
        float inv[9];
        load_face_geo(f, S, g, inv);
        L.g[0][slot] = g.x0; L.g[1][slot] = g.y0; L.g[2][slot] = g.x1; L.g[3][slot] = g.y1; L.g[4][slot] = g.x2; L.g[5][slot] = g.y2;
        L.g[6][slot] = g.z0; L.g[7][slot] = g.z1; L.g[8][slot] = g.z2;
#pragma unroll
        for (int k = 0; k < 9; k++) L.g[9 + k][slot] = inv[k];
        const int b = i / F;
        L.x_lo[slot] = cd.x_lo; L.y_lo[slot] = cd.y_lo; L.bw[slot] = cd.bw; L.img[slot] = b; L.fn[slot] = i - b * F;
    }
    const int row_end = wave_inclusive_sum(bh), row0 = row_end - bh;
    const int n_rows = __builtin_amdgcn_readlane(row_end, 63);
    constexpr bool pow2 = POW2;  // (S & (S - 1)) == 0: pixel centres with one multiply, no division in the search loop
    const float inv_s = 1.0f / (float)S;
    for (int rbase = 0; rbase < n_rows; rbase += FR_ROWS) {
        wave_lds_sync();  // (the previous window's readers are done)
        for (int r = max(rbase - row0, 0), r_hi = min(rbase + FR_ROWS - row0, bh); r < r_hi; ++r)
            L.rows[row0 + r - rbase] = (slot << 8) | r;
        wave_lds_sync();
        const int rows_here = min(FR_ROWS, n_rows - rbase);
        for (int rp = 0; rp < rows_here; rp += 64) {
            // 2. the inside interval [xa, xa + cnt) of this lane's row
            int item = 0, xa = 0, cnt = 0;
            if (rp + lane < rows_here) {
                item = L.rows[rp + lane];
                const int s = item >> 8;
                FaceGeo q;
                q.x0 = L.g[0][s]; q.y0 = L.g[1][s]; q.x1 = L.g[2][s]; q.y1 = L.g[3][s]; q.x2 = L.g[4][s]; q.y2 = L.g[5][s];
                const int x_lo = L.x_lo[s], x_end = x_lo + L.bw[s];
                const float yp = pixel_center_p(L.y_lo[s] + (item & 255), S, inv_s, pow2);
                // four pixels per step (independent tests overlap their latencies); the interval ends at the first outside
                // pixel behind an inside one
                bool started = false;
                for (int px = x_lo; px < x_end; px += 4) {
                    bool in[4];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        in[k] = inside_edges(q, pixel_center_p(px + k, S, inv_s, pow2), yp) && px + k < x_end;
                    const int m = (int)in[0] | ((int)in[1] << 1) | ((int)in[2] << 2) | ((int)in[3] << 3);
                    if (m && !started) {
                        xa = px + __ffs(m) - 1;
                        started = true;
                    }
                    cnt += __popc(m);
                    if (started && !in[3]) break;
                }
                xa -= x_lo;
            }
            const int pix_end = wave_inclusive_sum(cnt), pix0 = pix_end - cnt;
            const int n_pix = __builtin_amdgcn_readlane(pix_end, 63);
            for (int pbase = 0; pbase < n_pix; pbase += FR_PIX) {
                wave_lds_sync();
                for (int k = max(pbase - pix0, 0), k_hi = min(pbase + FR_PIX - pix0, cnt); k < k_hi; ++k)
                    L.pix[pix0 + k - pbase] = (item << 8) | (xa + k);  // slot, row, column
                wave_lds_sync();
                // 3. one inside pixel per lane
                const int pix_here = min(FR_PIX, n_pix - pbase);
                for (int pp = lane; pp < pix_here; pp += 64) {
                    unsigned long long key, *at;
                    if (slot_pixel(L, L.pix[pp], S, near_d, far_d, zbuf, key, at, epoch)) {
                        atomicMin(at, key);
                        if (touched) touched[(size_t)(at - zbuf) >> 6] = (unsigned char)epoch;  // (see k_resolve)
                    }
                }
            }
        }
    }
}

// Candidate pixels first + step * j (j = 0, 1, ...) of one face for the calling wave, lane by lane: the half-plane tests run on
// every candidate, the pixels that pass are collected (ballot-compacted) in the wave's LDS queue, and whenever 64 are waiting
// all lanes evaluate one each -- the divisions of the K2 body run with full lanes although only a fraction of a box (a few
// percent for a needle) lies inside its triangle.
constexpr int CQ = 128;  // queue words per wave: < 64 waiting + <= 64 new
template <class PixelOf>
__device__ __forceinline__ void raster_candidates(const FaceGeo &g, unsigned fnu, int n_cand, int first, int step,
                                                  PixelOf pixel_of, int *__restrict__ queue, int S, double near_d,
                                                  double far_d, unsigned long long *__restrict__ zimg, int epoch,
                                                  unsigned char *__restrict__ touched, size_t img_off)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const bool pow2 = (S & (S - 1)) == 0;
    const float inv_s = 1.0f / (float)S;
    int waiting = 0;
    auto evaluate = [&](int e) {
        const int x = e & 0xffff, y = e >> 16;
        float zp, w0, w1, w2;
        if (eval_inside(g, (float)x, (float)y, near_d, far_d, zp, w0, w1, w2)) {
            atomicMin(zimg + (size_t)y * S + x, zword(zp, fnu, epoch));
            if (touched) touched[(img_off + (size_t)y * S + x) >> 6] = (unsigned char)epoch;
        }
    };
    for (int base = first; base < n_cand; base += step) {  // (wave-uniform trip count)
        const int k = base + lane;
        int x = 0, y = 0;
        const bool in = k < n_cand && pixel_of(k, x, y) &&
                        inside_edges(g, pixel_center_p(x, S, inv_s, pow2), pixel_center_p(y, S, inv_s, pow2));
        const unsigned long long m = __ballot(in);
        if (in) queue[waiting + __popcll(m & below)] = (y << 16) | x;
        waiting += __popcll(m);
        if (waiting >= 64) {
            wave_lds_sync();
            const int e = queue[lane];
            const int rest = lane + 64 < waiting ? queue[lane + 64] : 0;
            wave_lds_sync();
            waiting -= 64;
            if (lane < waiting) queue[lane] = rest;
            evaluate(e);
        }
    }
    wave_lds_sync();
    if (lane < waiting) evaluate(queue[lane]);
    wave_lds_sync();  // (the queue is reused for the next face)
}

// Faces whose box is too large for k_face_raster and too small for a workgroup (a mesh of spiky or close-up triangles: config 4
// queues 1/5 of its faces): one wave per face, lanes stride over the box.
// Entry j of shard s is taken by consumer (j + s * units / QSHARDS) mod units -- the shards' entries start at different
// consumers, so that short shards do not all land on the first ones -- i.e. consumer u takes the entries j = u - start (mod
// units), j += units.
__device__ __forceinline__ int shard_first(int u, int s, int units, int nshards)
{
    const int start = (s * units) / nshards;  // (< units; units <= 2^13 consumers: no overflow)
    return u >= start ? u - start : u - start + units;
}

__device__ __forceinline__ void wave_raster(const float *__restrict__ faces, unsigned long long *__restrict__ zbuf,
                                            const int *__restrict__ wave_list, const int cnt, int qcap, int nshards, int F, int S,
                                            double near_d, double far_d, int *__restrict__ queue, int epoch,
                                            unsigned char *__restrict__ touched)
{
    const int waves = gridDim.x * (blockDim.x >> 6), u = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int s = 0; s < nshards; ++s) {
        const int n = __builtin_amdgcn_readlane(cnt, 2 * s + 1);
        for (int j = shard_first(u, s, waves, nshards); j < n; j += waves) {
            const int i = wave_list[(size_t)s * qcap + j];
            const float *f = faces + (size_t)i * 9;
            FaceGeo g;
            float inv[9];
            load_face_geo(f, S, g, inv);
            const Cand cd = face_candidates(g.x0, g.y0, g.x1, g.y1, g.x2, g.y2, S);
            const int b = i / F;
            raster_candidates(
                g, (unsigned)(i - b * F), cd.n, 0, 64,
                [&](int k, int &x, int &y) {
                    const int yy = k / cd.bw;
                    x = cd.x_lo + (k - yy * cd.bw);
                    y = cd.y_lo + yy;
                    return true;
                },
                queue, S, near_d, far_d, zbuf + (size_t)b * S * S, epoch, touched, (size_t)b * S * S);
        }
    }
}

// The two queues share one launch: a workgroup first takes large faces (its four waves on one face, 256 candidates per step),
// then its waves take medium ones.
__global__ __launch_bounds__(256) void k_large_raster(const float *__restrict__ faces,
                                                      unsigned long long *__restrict__ zbuf,
                                                      const int *__restrict__ large_list, const int *__restrict__ wave_list,
                                                      const int *__restrict__ n_large, int qcap, int nshards, int F, int S,
                                                      double near_d, double far_d, int epoch, unsigned char *__restrict__ touched)
{
    __shared__ int s_queue[4][CQ];
    __shared__ int s_cnt[2 * QSHARDS];  // entries of each shard's two queues (the counters start at -1)
    int *queue = s_queue[threadIdx.x >> 6];
    int any = 0;
    if (threadIdx.x < 2 * QSHARDS) {
        any = s_cnt[threadIdx.x] = (int)threadIdx.x < 2 * nshards ? n_large[(threadIdx.x >> 1) * QSTRIDE + (threadIdx.x & 1)] + 1 : 0;
    }
    if (!__syncthreads_or(any)) return;  // (a fine mesh: both queues empty)
    const int cnt = s_cnt[threadIdx.x & (2 * QSHARDS - 1)];  // lane 2 s / 2 s + 1: the entries of shard s's two queues
    for (int s = 0; s < nshards; ++s) {
        const int n = __builtin_amdgcn_readlane(cnt, 2 * s);
        for (int j = shard_first((int)blockIdx.x, s, (int)gridDim.x, nshards); j < n; j += gridDim.x) {
            const int i = large_list[(size_t)s * qcap + j];
            const float *f = faces + (size_t)i * 9;
            FaceGeo g;
            float inv[9];
            load_face_geo(f, S, g, inv);
            const Cand cd = face_candidates(g.x0, g.y0, g.x1, g.y1, g.x2, g.y2, S);
            const int b = i / F;
            raster_candidates(
                g, (unsigned)(i - b * F), cd.n, (int)(threadIdx.x >> 6) * 64, 256,
                [&](int k, int &x, int &y) { return cand_pixel(cd, k, S, x, y); }, queue, S, near_d, far_d,
                zbuf + (size_t)b * S * S, epoch, touched, (size_t)b * S * S);
        }
    }
    wave_raster(faces, zbuf, wave_list, cnt, qcap, nshards, F, S, near_d, far_d, queue, epoch, touched);
}

// --------------------------------------------------------------------------------------------------
// F3: shading, one pixel per thread (linear pixel index: fully coalesced map traffic).
// Shading of one pixel (K4 + K5, rasterize.py:361-465): shared by k_shade and the fused k_resolve.
__device__ __forceinline__ void shade_pixel(size_t i, int b, int fi, float w0, float w1, float w2, float depth,
                                            const float *__restrict__ faces, const float *__restrict__ zbase,
                                            const float *__restrict__ textures, float *__restrict__ rgb_map,
                                            int32_t *__restrict__ sampling_index_map,
                                            float *__restrict__ sampling_weight_map,
                                            const float *__restrict__ background, int bg_per_batch,
                                            float *__restrict__ alpha_map, int F, int ts, double eps, int fix_batch_z,
                                            const FaceLight &lit)
{
    if (alpha_map) alpha_map[i] = (fi >= 0) ? 1.0f : 0.0f;  // :449
    if (!rgb_map) return;
    float rgb[3];
    Taps t;
    if (fi >= 0) {
        // :389 (Q1): the reference reads batch element 0's geometry here; zbase = that element's faces (of the GLOBAL batch)
        const float *face = (fix_batch_z ? faces + (size_t)b * F * 9 : zbase) + (size_t)fi * 9;
        const float *texture = textures + ((size_t)b * F + fi) * ts * ts * ts * 3;   // :390
        bool flip = false;
        if (lit.light) {  // the cube of the original face; its reversed copy reads it transposed (nr_device.h: FaceLight)
            flip = fi >= lit.tex_faces;
            texture = textures + ((size_t)b * lit.tex_faces + (flip ? fi - lit.tex_faces : fi)) * ts * ts * ts * 3;
        }
        const float w[3] = {w0, w1, w2};
        const float fz[3] = {face[2], face[5], face[8]};
        compute_taps(fz, w, depth, ts, eps, t, flip);
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            if (t.isc[pn] >= ts * ts * ts) continue;  // outside the cube: weight 0 (see compute_taps), never dereferenced
            const float *tx = texture + t.isc[pn] * 3;
            rgb[0] += t.w[pn] * tx[0];
            rgb[1] += t.w[pn] * tx[1];
            rgb[2] += t.w[pn] * tx[2];
        }
        if (lit.light) {  // lighting.py:50-51 applied to the sample instead of to every texel
            const float *lc = lit.light + ((size_t)b * F + fi) * 3;
            rgb[0] *= lc[0];
            rgb[1] *= lc[1];
            rgb[2] *= lc[2];
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

__global__ __launch_bounds__(256) void k_shade(const float *__restrict__ faces, const float *__restrict__ zbase,
                                               const float *__restrict__ textures,
                                               const int32_t *__restrict__ face_index_map,
                                               const float *__restrict__ weight_map,
                                               const float *__restrict__ depth_map, float *__restrict__ rgb_map,
                                               int32_t *__restrict__ sampling_index_map,
                                               float *__restrict__ sampling_weight_map,
                                               const float *__restrict__ background, int bg_per_batch,
                                               float *__restrict__ alpha_map, int F, int S, int ts, double eps,
                                               int fix_batch_z, size_t n_pixels, FaceLight lit)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const int fi = face_index_map[i];
    const int b = (int)(i / ((size_t)S * S));
    float w0 = 0.0f, w1 = 0.0f, w2 = 0.0f, depth = 0.0f;
    if (rgb_map && fi >= 0) { w0 = weight_map[3 * i]; w1 = weight_map[3 * i + 1]; w2 = weight_map[3 * i + 2]; depth = depth_map[i]; }
    shade_pixel(i, b, fi, w0, w1, w2, depth, faces, zbase, textures, rgb_map, sampling_index_map, sampling_weight_map,
                background, bg_per_batch, alpha_map, F, ts, eps, fix_batch_z, lit);
}

// Everything the resolve pass needs (passed by value: one kernel argument block for its two launch shapes).
struct ResolveArgs {
    const float *faces;
    const unsigned long long *zbuf;
    int32_t *face_index_map;
    float *weight_map, *depth_map, *face_inv_map;
    unsigned char *visible_faces;
    int F, S;
    double near_d, far_d;
    size_t n_pixels;
    // fused shading (all NULL / 0 when not requested)
    const float *zbase, *textures;
    float *rgb_map;
    const float *background;
    int bg_per_batch;
    float *alpha_map;
    int ts;
    double eps;
    int fix_batch_z, epoch;
    int *queue_counters;
    FaceLight lit;
    int sparse_weights;
    const unsigned char *touched;
};

// One pixel of the resolve pass: decode the winner of z-buffer word i (`drawn` false: nobody drew near it, the word is not
// read), re-evaluate it exactly as the candidate tests did, write the maps and shade.
__device__ __forceinline__ void resolve_pixel(const ResolveArgs &a, size_t i, bool drawn)
{
    const float *__restrict__ faces = a.faces;
    const int S = a.S, F = a.F, epoch = a.epoch;
    const unsigned long long pk = drawn ? a.zbuf[i] : ZEMPTY;
    int fn = -1;
    float zp = (float)a.far_d, w0 = 0.0f, w1 = 0.0f, w2 = 0.0f;  // rasterize.py:296, :478-480
    float inv[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    int b = 0;
    if (a.rgb_map || a.alpha_map) b = (int)(i / ((size_t)S * S));
    const bool hit = drawn && (epoch < 0 ? pk != ZEMPTY : (int)(pk >> 56) == epoch);
    if (hit) {
        fn = epoch < 0 ? (int)(unsigned)(pk & 0xffffffffu) : (int)(unsigned)(pk & 0xffffffu);
        const size_t SS = (size_t)S * S;
        b = (int)(i / SS);
        const int pn = (int)(i - (size_t)b * SS);
        const int py = pn / S, px = pn - py * S;
        FaceGeo g;
        load_face_geo(faces + ((size_t)b * F + fn) * 9, S, g, inv);
        eval_pixel(g, pixel_center_f(px, S), pixel_center_f(py, S), (float)px, (float)py, a.near_d, a.far_d, zp, w0, w1, w2);
        if (a.visible_faces) a.visible_faces[(size_t)b * F + fn] = 1;  // same value from every pixel of the face: no atomic
    }
    a.face_index_map[i] = fn;
    if (a.depth_map) a.depth_map[i] = zp;
    // (NR_FLAG_SPARSE_WEIGHT_MAP: the zeros of the pixels no face covers -- 7 of 8 on a teapot view, 44 of the 50 MB of this
    // map at the headline size -- are not stored; the backward reads weights of covered pixels only)
    if (a.weight_map && (hit || !a.sparse_weights)) {
        float *w = a.weight_map + 3 * i;
        w[0] = w0;
        w[1] = w1;
        w[2] = w2;
    }
    if (a.face_inv_map) {
        float *o = a.face_inv_map + 9 * i;
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = inv[k];
    }
    if (a.rgb_map || a.alpha_map)
        shade_pixel(i, b, fn, w0, w1, w2, zp, faces, a.zbase, a.textures, a.rgb_map, nullptr, nullptr, a.background,
                    a.bg_per_batch, a.alpha_map, F, a.ts, a.eps, a.fix_batch_z, a.lit);
}

// epoch mode: nobody fills the workspace for the next call, so the queue counters go back to -1 in the resolve pass (the
// raster kernels that read them are done: this launch is behind them on the stream)
__device__ __forceinline__ void reset_queue_counters(const ResolveArgs &a)
{
    if (blockIdx.x == 0 && threadIdx.x < QSHARDS && a.epoch >= 0) {
        a.queue_counters[threadIdx.x * QSTRIDE] = -1;
        a.queue_counters[threadIdx.x * QSTRIDE + 1] = -1;
    }
}

__global__ __launch_bounds__(256) void k_resolve(ResolveArgs a)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    reset_queue_counters(a);
    if (i >= a.n_pixels) return;
    // Epoch mode keeps a byte per 64 consecutive pixels (= the pixels of one wave here) that every z-buffer update of this call
    // sets to the call's epoch number: where it holds anything else nobody drew -- 7 of 8 segments of a teapot view -- and the
    // 512 bytes of z-buffer behind it are not read (round 4: 33.5 -> ~6 MB of z-buffer reads at the headline size).  Stale
    // bytes of earlier calls carry larger epoch numbers, the initial fill 0xff: no clearing.
    resolve_pixel(a, i, !a.touched || a.touched[i >> 6] == (unsigned char)a.epoch);
}

// The same pass for epoch mode on rasters with an even side.  Most of what the pass writes is the constant of undrawn pixels
// -- 88 % of the 108 MB at the headline size -- and one pixel per lane stores those as dwords (the RGB ones 12 bytes apart).
// Here the first wave of a workgroup fills the undrawn ones of the workgroup's four segments FOUR consecutive pixels per lane,
// with 16-byte stores (a quad never straddles segments or images: 4 | 64, 4 | S * S); the drawn segments are resolved one
// pixel per lane as before, and the other waves of an undrawn stretch leave at once.  (Fused forward of the headline batch
// 72.3 -> 66.2 us; workgroups of 512 / 1024 pixels with 2 / 4 passes per lane: 72.9 / 82.6, profiles/r04_fwd_variants.jsonl.)
// (256 pixels per workgroup: 64 / 128 / 512 / 1024 were measured -- 74.7 / 73.7 / 76.5 / 84.5 us against 69-70)
__global__ __launch_bounds__(256) void k_resolve_quads(ResolveArgs a)
{
    reset_queue_counters(a);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, i0 = (size_t)blockIdx.x * 256 + 4 * threadIdx.x;
    const unsigned char ep = (unsigned char)a.epoch;
    unsigned char tq = ep, ti = (unsigned char)(ep + 1);
    if (threadIdx.x < 64 && i0 < a.n_pixels) tq = a.touched[i0 >> 6];
    if (i < a.n_pixels) ti = a.touched[i >> 6];
    if (tq != ep) {
        reinterpret_cast<int4 *>(a.face_index_map)[i0 >> 2] = make_int4(-1, -1, -1, -1);
        const float zf = (float)a.far_d;  // rasterize.py:296
        if (a.depth_map) reinterpret_cast<float4 *>(a.depth_map)[i0 >> 2] = make_float4(zf, zf, zf, zf);
        const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (a.weight_map && !a.sparse_weights) {
            float4 *w = reinterpret_cast<float4 *>(a.weight_map + 3 * i0);
            w[0] = zero; w[1] = zero; w[2] = zero;
        }
        if (a.face_inv_map) {
            float4 *o = reinterpret_cast<float4 *>(a.face_inv_map + 9 * i0);
#pragma unroll
            for (int k = 0; k < 9; k++) o[k] = zero;
        }
        if (a.alpha_map) reinterpret_cast<float4 *>(a.alpha_map)[i0 >> 2] = zero;  // :449
        if (a.rgb_map) {
            const float *bg = a.background + (a.bg_per_batch ? 3 * (int)(i0 / ((size_t)a.S * a.S)) : 0);
            float c[3];
#pragma unroll
            for (int k = 0; k < 3; k++) c[k] = 0.0f * 0.0f + 1.0f * bg[k];  // :463 with mask = 0 (shade_pixel)
            float4 *o = reinterpret_cast<float4 *>(a.rgb_map + 3 * i0);
            o[0] = make_float4(c[0], c[1], c[2], c[0]);
            o[1] = make_float4(c[1], c[2], c[0], c[1]);
            o[2] = make_float4(c[2], c[0], c[1], c[2]);
        }
    }
    if (ti == ep) resolve_pixel(a, i, true);
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
        case NR_E_INDEX: return "nr: a vertex index lies outside [0, num_vertices)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "nr: unknown error";
    }
}

namespace {
struct FwdLayout {
    size_t zbuf_off, list_off, count_off, touch_off, total;
};
// entries a queue shard can receive: the faces of the raster workgroups blockIdx = shard (mod nshards), 4 x faces_per_wave each
size_t queue_shard_capacity(size_t n_faces, int faces_per_wave, int nshards)
{
    const size_t per_wg = (size_t)4 * faces_per_wave, n_wg = (n_faces + per_wg - 1) / per_wg;
    return (n_wg + nshards - 1) / nshards * per_wg;
}

FwdLayout fwd_layout(int B, int F, int S)
{
    FwdLayout L;
    const size_t n = (size_t)B * F, P = (size_t)B * S * S;
    L.zbuf_off = 0;
    L.count_off = L.zbuf_off + P * sizeof(unsigned long long);  // the queue counters sit right behind the z-buffer (one fill)
    L.list_off = align_up(L.count_off + (size_t)QSHARDS * QSTRIDE * sizeof(int), 256);
    // the queue of large faces, then the queue of medium ones: QSHARDS shards each, a shard as long as what the raster
    // workgroups that feed it hold (queue_shard_capacity; over all shards at most n + QSHARDS * 256 entries)
    L.touch_off = align_up(L.list_off + 2 * (n + (size_t)QSHARDS * 256) * sizeof(int), 256);
    L.total = L.touch_off + (P + 63) / 64;  // epoch mode: one byte per 64 pixels, "drawn in this call" (k_resolve)
    return L;
}
}  // namespace

NR_API size_t nr_forward_workspace_bytes(int32_t B, int32_t F, int32_t S)
{
    if (check_sizes(B, F, S)) return 0;
    return fwd_layout(B, F, S).total;
}

namespace {
int run_forward(const float *faces, int32_t *face_index_map, float *weight_map, float *depth_map, float *face_inv_map,
                unsigned char *visible_faces, int B, int F, int S, double near, double far, void *workspace,
                size_t workspace_bytes, hipStream_t st, const float *faces_z_ref, const float *textures, float *rgb_map,
                const float *background, int bg_per_batch, float *alpha_map, int ts, double eps, int fix_batch_z,
                int flags = 0, const FaceLight &lit = FaceLight())
{
    if (!faces || !face_index_map) return NR_E_NULL;
    if (int e = check_sizes(B, F, S)) return e;
    const FwdLayout L = fwd_layout(B, F, S);
    if (!workspace || workspace_bytes < L.total) return NR_E_WORKSPACE;
    const size_t n = (size_t)B * F, P = (size_t)B * S * S;
    unsigned char *ws = (unsigned char *)workspace;
    unsigned long long *zbuf = (unsigned long long *)(ws + L.zbuf_off);
    int *n_large = (int *)(ws + L.count_off);
    int *large_list = (int *)(ws + L.list_off);

    // Epoch mode (NR_FLAG_ZBUF_EPOCH, see zword): the caller keeps the workspace, filled it with 0xff bytes once and counts
    // the epoch down; nothing is filled here.  It needs face indices below 2^24; otherwise, and without the flag:
    // one fill: ZEMPTY words and, right behind them, the queue counters at -1
    int epoch = -1;
    if ((flags & NR_FLAG_ZBUF_EPOCH) && F < (1 << 24)) {
        epoch = (flags >> 8) & 0xff;
        if (epoch > 254) return NR_E_MODE;  // 255 is the epoch of a freshly filled word
    } else {
        if (int he = fill_bytes(zbuf, 0xff, P * sizeof(unsigned long long) + (size_t)QSHARDS * QSTRIDE * sizeof(int), st)) return he;  // (nr_device.h: not a memset node)
    }
    const bool small_launch = n < SMALL_LAUNCH_FACES;
    // (launches of the 16-faces-per-wave kernel keep one shard: fine meshes queue a few faces, and a consumer that walks 16
    // nearly empty shards costs the headline 2.4 us)
    const int nshards = small_launch ? 1 : QSHARDS;
    const int qcap = (int)queue_shard_capacity(n, small_launch ? 16 : 64, nshards);
    int *wave_list = large_list + (size_t)nshards * qcap;
    unsigned char *touched = epoch >= 0 ? ws + L.touch_off : nullptr;
    {
        const bool pow2 = (S & (S - 1)) == 0, small = small_launch;
#define NR_FACE_RASTER(P, FC, G)                                                                                           \
    hipLaunchKernelGGL((k_face_raster<P, FC, G>), dim3((unsigned)((n + 4 * FC - 1) / (4 * FC))), dim3(256), 0, st, faces, zbuf, \
                       large_list, wave_list, n_large, qcap, nshards, visible_faces, (int)n, F, S, near, far, epoch, touched)
        if (small) { if (pow2) NR_FACE_RASTER(true, 16, 4); else NR_FACE_RASTER(false, 16, 4); }
        else { if (pow2) NR_FACE_RASTER(true, 64, 16); else NR_FACE_RASTER(false, 64, 16); }
#undef NR_FACE_RASTER
    }
    // a resident grid loops over the two queues; with empty queues (a fine mesh) its workgroups read two counters and leave --
    // which costs what dispatching them costs (2048 workgroups: 4.7 us), so small launches get a smaller grid (a face per 256
    // of the call's, 256 .. 2048 workgroups: the queues hold a fraction of the faces, and each large face is a workgroup's work)
    const unsigned queue_wgs = (unsigned)(n / 256 < 256 ? 256 : (n / 256 > 2048 ? 2048 : n / 256));
    hipLaunchKernelGGL(k_large_raster, dim3(queue_wgs), dim3(256), 0, st, faces, zbuf, large_list, wave_list, n_large, qcap, nshards,
                       F, S, near, far, epoch, touched);
    ResolveArgs ra;
    ra.faces = faces; ra.zbuf = zbuf; ra.face_index_map = face_index_map; ra.weight_map = weight_map; ra.depth_map = depth_map;
    ra.face_inv_map = face_inv_map; ra.visible_faces = visible_faces; ra.F = F; ra.S = S; ra.near_d = near; ra.far_d = far;
    ra.n_pixels = P; ra.zbase = faces_z_ref ? faces_z_ref : faces; ra.textures = textures; ra.rgb_map = rgb_map;
    ra.background = background; ra.bg_per_batch = bg_per_batch; ra.alpha_map = alpha_map; ra.ts = ts; ra.eps = eps;
    ra.fix_batch_z = fix_batch_z; ra.epoch = epoch; ra.queue_counters = n_large; ra.lit = lit;
    ra.sparse_weights = (flags & NR_FLAG_SPARSE_WEIGHT_MAP) ? 1 : 0; ra.touched = touched;
    const uintptr_t align = (uintptr_t)face_index_map | (uintptr_t)weight_map | (uintptr_t)depth_map | (uintptr_t)face_inv_map |
                            (uintptr_t)rgb_map | (uintptr_t)alpha_map;
    if (touched && S % 2 == 0 && (align & 15) == 0)  // (k_resolve_quads: 16-byte stores into every map)
        hipLaunchKernelGGL(k_resolve_quads, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, ra);
    else
        hipLaunchKernelGGL(k_resolve, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, ra);
    return launch_status();
}
}  // namespace

NR_API int nr_forward_face_index_map(const float *faces, int32_t *face_index_map, float *weight_map, float *depth_map,
                                     float *face_inv_map, uint8_t *visible_faces, int32_t B, int32_t F, int32_t S,
                                     double near, double far, void *workspace, size_t workspace_bytes, void *stream)
{
    return run_forward(faces, face_index_map, weight_map, depth_map, face_inv_map, visible_faces, B, F, S, near, far,
                       workspace, workspace_bytes, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0,
                       0.0, 0);
}

// Fused forward = Rasterize.forward_gpu (rasterize.py:467-513): visibility, texture sampling, background and alpha
// behind one entry point; the shading runs inside the resolve pass, on the winner still in registers.
NR_API int nr_forward_rasterize(const float *faces, const float *faces_z_ref, const float *textures,
                                int32_t *face_index_map, float *weight_map, float *depth_map, float *rgb_map,
                                float *alpha_map, uint8_t *visible_faces, const float *background, int32_t bg_per_batch,
                                int32_t B, int32_t F, int32_t S, int32_t ts, double near, double far, double eps,
                                int32_t flags, void *workspace, size_t workspace_bytes, void *stream)
{
    return nr_forward_rasterize_lit(nullptr, faces, faces_z_ref, textures, face_index_map, weight_map, depth_map, rgb_map,
                                    alpha_map, visible_faces, background, bg_per_batch, B, F, S, ts, near, far, eps, flags,
                                    workspace, workspace_bytes, stream);
}

// host-side check of an nr_face_light (see include/nr_hip.h) -> the kernels' FaceLight
int nr::face_light_args(const nr_face_light *lit, int F, bool backward, FaceLight &out)
{
    out = FaceLight();
    if (!lit) return 0;
    if (!lit->light) return NR_E_NULL;
    if (lit->texture_faces < 1 || (lit->texture_faces != F && 2 * (int64_t)lit->texture_faces != F)) return NR_E_SIZE;
    out.light = lit->light;
    out.tex_faces = lit->texture_faces;
    if (backward) {
        out.textures = lit->textures;
        out.grad_light = lit->grad_light;
        if (out.grad_light && !out.textures) return NR_E_NULL;
    }
    return 0;
}

NR_API int nr_forward_rasterize_lit(const nr_face_light *lit, const float *faces, const float *faces_z_ref,
                                    const float *textures, int32_t *face_index_map, float *weight_map, float *depth_map,
                                    float *rgb_map, float *alpha_map, uint8_t *visible_faces, const float *background,
                                    int32_t bg_per_batch, int32_t B, int32_t F, int32_t S, int32_t ts, double near,
                                    double far, double eps, int32_t flags, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    if (rgb_map) {
        if (!textures || !background) return NR_E_NULL;
        if (ts < 2 || ts > 1024) return NR_E_SIZE;
    }
    FaceLight fl;
    if (int e = face_light_args(rgb_map ? lit : nullptr, F, false, fl)) return e;
    return run_forward(faces, face_index_map, weight_map, depth_map, nullptr, visible_faces, B, F, S, near, far,
                       workspace, workspace_bytes, (hipStream_t)stream, faces_z_ref, textures, rgb_map, background,
                       bg_per_batch, alpha_map, ts, eps, (flags & NR_FLAG_FIX_TEXTURE_BATCH_Z) ? 1 : 0, flags, fl);
}

NR_API int nr_forward_texture_sampling(const float *faces, const float *faces_z_ref, const float *textures,
                                       const int32_t *face_index_map, const float *weight_map, const float *depth_map,
                                       float *rgb_map, int32_t *sampling_index_map, float *sampling_weight_map,
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
    hipLaunchKernelGGL(k_shade, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, faces,
                       faces_z_ref ? faces_z_ref : faces, textures, face_index_map, weight_map, depth_map, rgb_map,
                       sampling_index_map, sampling_weight_map, background, bg_per_batch, alpha_map, F, S, ts, eps,
                       (flags & NR_FLAG_FIX_TEXTURE_BATCH_Z) ? 1 : 0, n, FaceLight());
    return launch_status();
}
