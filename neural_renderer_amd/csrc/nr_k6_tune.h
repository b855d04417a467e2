// nr_k6_tune.h -- the numerics / shape knobs of K6's default (tolerance-mode) kernel in one place.
//
// The product build takes the defaults below.  Development builds (neural_renderer_amd._build.build_variant, timed side
// by side through NR_HIP_LIB) override single knobs with -D...; nothing else in the library is conditional on macros.
// What each knob costs and buys is measured in profiles/r04_k6_numerics.jsonl (LAB-NOTEBOOK, round 4).
#pragma once

#ifndef NR_K6_NEWTON       // one Newton step on each v_rcp_f32 of a visit (2 fma): the reciprocal to ~0.5 ulp instead of 1
#define NR_K6_NEWTON 0
#endif
#ifndef NR_K6_FUSED_DIFF   // diff = sum (I - ref) * g accumulated with fused multiply-adds (0: the reference's roundings)
#define NR_K6_FUSED_DIFF 1
#endif
#ifndef NR_K6_FUSED_DIST   // dist = fma(c, t, +-eps) (0: c * t rounded, then +- eps: the reference's two roundings)
#define NR_K6_FUSED_DIST 1
#endif
#ifndef NR_K6_BATCH_DOUBLE // the float sums of a batch of NR_K6_FB visits are added to DOUBLE piece sums (0: float piece sums)
#define NR_K6_BATCH_DOUBLE 0
#endif
#ifndef NR_K6_RUNSUM_DOUBLE  // the DPP run sums in front of the LDS atomics in double (0: float)
#define NR_K6_RUNSUM_DOUBLE 0
#endif
#ifndef NR_K6_FB           // pixels of an unrolled piece whose LDS reads are requested together (FSEG is a multiple;
#define NR_K6_FB 3         // 1 / 3 / 5 -> stage 229 / 230 / 252 us, 3 needs the fewest registers)
#endif
#ifndef NR_K6_U_GROUP      // class U pieces per super-piece (one descriptor / decode / flush for up to this many pieces of 15 pixels;
#define NR_K6_U_GROUP 3      // 1: every piece on its own, rounds 3-4; at most 4: the count travels in two bits)
#endif
#ifndef NR_K6_SMALL_RASTER_MAX  // up to this raster size: 256-thread workgroups on two-line bands (band_shape; 0: never)
#define NR_K6_SMALL_RASTER_MAX 400
#endif
#ifndef NR_K6_MINWAVES_256  // launch bound (waves per SIMD) of the 256-thread shape
#define NR_K6_MINWAVES_256 4
#endif
#ifndef NR_K6_WMAX         // widest band (lines per workgroup) of the 512-thread shape
#define NR_K6_WMAX 4
#endif
#ifndef NR_K6_FOLD_KB      // largest slice of the fused backward's grad_textures fill that a band workgroup takes along
#define NR_K6_FOLD_KB 128
#endif
#ifndef NR_K6_LDS_BUDGET   // (512-thread shape) three workgroups per 160 KB CU, allocation granules of 512 bytes included (3 x 53.5 KB would not fit)
#define NR_K6_LDS_BUDGET (53 * 1024)
#endif

#ifndef NR_K6_WIDE_BUDGET_FROM  // rasters in (FROM, TO]: the 512-thread shape with 80 KB of LDS per workgroup (band_shape)
#define NR_K6_WIDE_BUDGET_FROM 576
#endif
#ifndef NR_K6_WIDE_BUDGET_TO
#define NR_K6_WIDE_BUDGET_TO 832
#endif
#ifndef NR_K6_OVF_GRID      // workgroups of k_bpm_fast's overflow-only launch behind k_bpm_row (images whose records exceed the line buffer):
#define NR_K6_OVF_GRID 256  // one per CU -- a launch whose workgroups leave at once costs 1.3 us up to 256 of them, 1.6 at 1024, 2.3 at
#endif                      // 4096, 4.6 at 16 384 (scripts/dev/empty_launch_probe.hip); fused backward at the headline shape 214.4 -> 212.5 us
#ifndef NR_ROW_MIN_WGS      // k_bpm_row: four-line bands become two-line bands while the launch has fewer band workgroups than this ...
#define NR_ROW_MIN_WGS 4096
#endif
#ifndef NR_ROW_MIN_WGS_1    // ... and two-line bands one-line bands below this many
#define NR_ROW_MIN_WGS_1 1024
#endif

#ifndef NR_SHARED_LAUNCH_MAX_FACES  // fused backward: calls of up to this many faces (batch x faces) put the line setup and the
#define NR_SHARED_LAUNCH_MAX_FACES 98304  // K7 / K8 gather into one launch (nr_backward_rasterize_lit; measured: LAB-NOTEBOOK, late round 4)
#endif

namespace nr {
namespace k6 {
constexpr bool NEWTON = NR_K6_NEWTON != 0;
constexpr bool FUSED_DIFF = NR_K6_FUSED_DIFF != 0;
constexpr bool FUSED_DIST = NR_K6_FUSED_DIST != 0;
constexpr bool BATCH_DOUBLE = NR_K6_BATCH_DOUBLE != 0;
constexpr bool RUNSUM_DOUBLE = NR_K6_RUNSUM_DOUBLE != 0;
constexpr int FB = NR_K6_FB;
constexpr int U_GROUP = NR_K6_U_GROUP;
static_assert(U_GROUP >= 1 && U_GROUP <= 4, "a super-piece's count travels in two bits");
constexpr int SMALL_RASTER_MAX = NR_K6_SMALL_RASTER_MAX;
constexpr int MINWAVES_256 = NR_K6_MINWAVES_256;
constexpr int WMAX = NR_K6_WMAX;
constexpr int FOLD_KB = NR_K6_FOLD_KB;
constexpr unsigned long LDS_BUDGET = NR_K6_LDS_BUDGET;
constexpr unsigned long ROW_MIN_WGS = NR_ROW_MIN_WGS, ROW_MIN_WGS_1 = NR_ROW_MIN_WGS_1;
constexpr unsigned OVF_GRID = NR_K6_OVF_GRID;
constexpr int WIDE_BUDGET_FROM = NR_K6_WIDE_BUDGET_FROM, WIDE_BUDGET_TO = NR_K6_WIDE_BUDGET_TO;
constexpr unsigned long SHARED_LAUNCH_MAX_FACES = NR_SHARED_LAUNCH_MAX_FACES;
}  // namespace k6
}  // namespace nr
