/*
 * nr_hip_profile.h -- the measurement hook of the MEASUREMENT build of the library (libnr_hip_prof.so: the same sources as
 * libnr_hip.so compiled with -DNR_PROFILE_HOOK, neural_renderer_amd._build.build_profile()).  The product library
 * (include/nr_hip.h) neither exports these functions nor keeps their state: up to 0.4.1 they were part of the product ABI, which
 * put a process-wide event pair into a library whose header promises "keeps no state between calls".
 * Used by bench.py (`roofline.avg_launch_us`: the dominant kernel's own duration, measured live with HIP events on the launch
 * stream) and tests/test_hip_parity.py::test_band_kernel_timing_hook.
 */
#ifndef NR_HIP_PROFILE_H
#define NR_HIP_PROFILE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* With enable != 0 every K6 band-kernel launch of nr_backward_pixel_map /
 * nr_backward_rasterize[_lit] is bracketed by a pair of HIP events recorded on the call's stream (events of the library's own,
 * created on the first enable); nr_profile_band_kernel_ms() waits for the last pair and returns the time between them in
 * milliseconds (< 0: no launch was bracketed, or an event call failed).  That is the duration of the path's dominant kernel
 * alone, without the helper launches of its stage call, as `rocprofv3 --kernel-trace --stats` reports it.  Process-wide, not
 * thread-safe, and the two event packets cost the stream a few microseconds per call: off outside measurements. */
int nr_profile_band_kernel(int32_t enable);
float nr_profile_band_kernel_ms(void);
/* which band kernel the last bracketed launch was: 0 k_bpm_fast, 1 k_bpm_row (-1: none) */
int nr_profile_band_kernel_which(void);
/* which band kernel a call takes, as a function of the call (host logic only, callable without a device): 1 k_bpm_row, 0
 * k_bpm_fast (NR_FLAG_K6_SCAN / _LEGACY, rasters beyond 1024, the default mode with eps = 0).  The batch size does not enter. */
int nr_profile_k6_choice(int32_t B, int32_t F, int32_t S, int32_t return_rgb, int32_t return_alpha, double eps, int32_t flags);

#ifdef __cplusplus
}
#endif
#endif /* NR_HIP_PROFILE_H */
