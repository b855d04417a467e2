/*
 * nr_hip.h -- C ABI of libnr_hip.so: the MI355X (gfx950) rasterizer hot path of neural_renderer.
 *
 * Drop-in boundary.  The reference launches its stages through
 *     chainer.cuda.elementwise(in_params, out_params, body, name)(loop, *arrays)
 * i.e. CuPy's ElementwiseKernel FFI: raw device arrays + constants pasted into the source, launched
 * asynchronously on the current stream.  Each entry point below replaces one such launch site and is
 * named after the reference method that owns it (reference = /root/reference/neural_renderer/rasterize.py):
 *
 *   nr_forward_face_index_map    <- Rasterize.forward_face_index_map_gpu   rasterize.py:94-359  (K1+K2)
 *   nr_forward_texture_sampling  <- Rasterize.forward_texture_sampling      rasterize.py:361-438 (K4)
 *                                   + forward_background_gpu / forward_alpha_map_gpu  :440-465   (K5)
 *   nr_backward_pixel_map        <- Rasterize.backward_pixel_map_gpu        rasterize.py:517-748 (K6)
 *   nr_backward_textures         <- Rasterize.backward_textures_gpu         rasterize.py:750-792 (K7)
 *   nr_backward_depth_map        <- Rasterize.backward_depth_map_gpu        rasterize.py:794-847 (K8)
 *   nr_forward_rasterize         <- Rasterize.forward_gpu  (K1+K2 -> K4+K5 fused) rasterize.py:467-513
 *   nr_backward_rasterize        <- Rasterize.backward_gpu (K6 -> K7 -> K8 fused)  rasterize.py:849-889
 *   nr_vertices_to_faces[_backward] <- vertices_to_faces + its get_item backward    vertices_to_faces.py:4-21
 *   nr_image_epilogue[_backward]  <- transpose + flip + average_pooling_2d of rasterize_rgbad   rasterize.py:953-969
 *   nr_adam_update                <- AdamRule.update_core_gpu                                  optimizers.py:17-34
 *   nr_load_textures              <- load_textures kernel of load_obj(load_texture=True)      load_obj.py:87-144
 *   nr_create_texture_image       <- create_texture_image kernels of save_obj(textures=...)   save_obj.py:32-146
 *   nr_frontend_forward/_backward <- fill_back + lighting + look_at/look + perspective + vertices_to_faces
 *                                    of Renderer.render*                                  renderer.py:35-107
 *   nr_forward_rasterize_lit / nr_backward_rasterize_lit / nr_frontend_forward_light / _backward_light: the fused
 *                                   rasterizer and front-end with per-face light colours instead of lit, duplicated
 *                                   textures (renderer.py:77-103 with lighting.py:50-51 moved into K4 / K7)
 *
 * Conventions
 *   - plain device pointers (hipMalloc / torch caching allocator memory), C-contiguous, float32 / int32, every buffer
 *     16-byte aligned (hipMalloc gives 256, torch 512: the kernels move maps, textures and workspaces as 16-byte words);
 *     sizes are int32; near / far / eps are doubles because the reference pastes their Python repr into
 *     the kernel source as double literals (rasterize.py:226-234, 428-433, 737-743).
 *   - the library owns no memory and keeps no state between calls (it reads no environment variable; the one thing it
 *     remembers is which dynamic-LDS limit the driver already granted to a kernel): outputs, residuals and scratch
 *     ("workspace", size from nr_*_workspace_bytes) all belong to the caller.
 *   - every launch goes to `stream` (a hipStream_t passed as void*; NULL = the null stream), is
 *     asynchronous, and never synchronises the device.  Functions are re-entrant and thread-safe.
 *   - return value: 0 on success; < 0 argument error (NR_E_*); > 0 a hipError_t from the launch.
 *     nr_error_string() renders either.
 *   - layouts (B batch, F faces, S raster size, ts texture size):
 *       faces [B,F,3,3] xyz per vertex: x,y in NDC (y up), z = positive camera depth
 *       textures / grad_textures [B,F,ts,ts,ts,3]
 *       face_index_map [B,S,S] int32 (-1 = no face), weight_map [B,S,S,3], depth_map [B,S,S],
 *       face_inv_map [B,S,S,3,3], rgb_map / grad_rgb_map [B,S,S,3], alpha_map / grad_alpha_map [B,S,S],
 *       sampling_index_map [B,S,S,8] int32, sampling_weight_map [B,S,S,8]; row 0 is the BOTTOM row
 *       (the public Python API flips afterwards, rasterize.py:956-960).
 */
#ifndef NR_HIP_H
#define NR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NR_VERSION 600 /* 0.6.0: K6's two arithmetic modes on ONE band kernel for every call size (k_bpm_row: a line record per 16 lanes,
                          *        the sums of a record on the matrix pipe in double); NR_FLAG_K6_PX is ignored;
                          * 0.5.0: K6's default mode on the lane-parallel band kernel (k_bpm_px; NR_FLAG_K6_LEGACY keeps k_bpm_fast); the
                          *        measurement hook nr_profile_band_kernel left the product ABI (include/nr_hip_profile.h, libnr_hip_prof.so);
                          * 0.4.1: NR_FLAG_SERIAL_BACKWARD (the fused backward's gather shares a launch with K6's line setup);
                          * 0.4.0: NR_FLAG_EXACT_GRADIENT and NR_FLAG_K6_SCAN combine (one band kernel, two arithmetic modes);
                          *        NR_FLAG_SPARSE_WEIGHT_MAP;
                          * 0.3.0: any `near` (NR_E_NEAR removed); 0.2.0: faces_z_ref, visible_faces, flags on the K6 entry points */

/* argument errors */
#define NR_E_NULL (-1)      /* a required pointer is NULL */
#define NR_E_SIZE (-2)      /* a size is out of range (B,F,S < 1, B > 65535, S > 16384, ts < 2, index overflow) */
#define NR_E_WORKSPACE (-3) /* workspace missing or too small */
#define NR_E_MODE (-4)      /* nothing to do / inconsistent optional arguments */
/* (-5 was NR_E_NEAR up to 0.2.0: `near` may be any number now, as in the reference, rasterize.py:331) */
#define NR_E_INDEX (-6)     /* a vertex index outside [0, num_vertices) */

/* flags */
#define NR_FLAG_FIX_TEXTURE_BATCH_Z 1 /* texture sampling (K4 / K7): read the face's z from the pixel's own batch element
                                         instead of batch 0 (the reference reads batch 0: rasterize.py:389) */
#define NR_FLAG_EXACT_GRADIENT 2      /* K6: every per-pixel term with the reference's arithmetic (its operations one by one, IEEE
                                         division, the double `dist +- eps`), all sums in double: bit-identical terms, bound 2e-6
                                         against the exactly summed reference terms.  Default (flag clear): float terms through
                                         fused multiply-adds and v_rcp_f32, ~1 ulp per term: within the north star's 1e-4 of the
                                         exactly summed reference terms (measured <= 5e-5 on the BASELINE configurations) -- plus,
                                         on entries that cancel down to 1e-3 of the largest gradient, where one ulp of a large
                                         term counts as 1.2e-4, twice the reference's own float-summation noise (worst scene of a
                                         soak built to cancel: 1.8e-4 where the reference's own sums are 7e-4 off).  Same sweep
                                         structure either way; what the modes cost and measure:
                                         profiles/<round>_parity_summary.md, DESIGN.md 3.
                                         REPRODUCIBILITY (since 0.6.0): both modes return the same grad_faces bits from call to
                                         call, for a batch and its shards, and for fused and staged calls wherever k_bpm_row runs
                                         (raster <= 1024): a record's sums have a fixed order and everything above is added in
                                         double.  NR_FLAG_K6_LEGACY / the scan path in the default mode (k_bpm_fast: float run sums
                                         grouped by arrival) agree to ~1.2e-5 of the largest gradient between two calls. */
#define NR_FLAG_K6_GLOBAL 4           /* K6: force the global-memory kernel that otherwise only serves rasters whose band
                                         does not fit in LDS (a testing aid) */
#define NR_FLAG_K6_SCAN 8             /* K6: let every band workgroup derive its lines from the image's visible-face list
                                         itself, the path that otherwise only serves images whose line records exceed the
                                         workspace's record buffer (a testing aid) */

#define NR_FLAG_ZBUF_EPOCH 16          /* nr_forward_rasterize: the forward workspace is KEPT by the caller between calls (same
                                         sizes, same stream order) and this call's epoch number is in bits 8..15 of `flags`
                                         (0..254): the z-buffer is neither filled before nor cleaned after the call -- a word
                                         written under a larger epoch number loses every atomic minimum against this call's
                                         and reads as empty.  Contract: fill the workspace with 0xff bytes once, then call with
                                         epochs 254, 253, ..., 0; fill again before starting over (or call without the flag,
                                         which fills).  Needs num_faces < 2^24 (ignored otherwise).  Saves the 8 B / pixel fill
                                         of every forward (7 us and 33.5 MB at the headline size). */
#define NR_ZBUF_EPOCH_FLAGS(e) (NR_FLAG_ZBUF_EPOCH | (((e) & 0xff) << 8))
#define NR_FLAG_SPARSE_WEIGHT_MAP 32   /* nr_forward_rasterize: weight_map is written for the pixels a face covers only; the
                                         elements of uncovered pixels (zeros in the reference, rasterize.py:479) are left as
                                         they are.  For callers that keep weight_map as a residual of the backward -- which
                                         reads it at covered pixels only -- and do not hand it out: 7 of 8 pixels of a teapot
                                         view are uncovered (44 of the map's 50 MB at the headline size). */
#define NR_FLAG_SERIAL_BACKWARD 64     /* nr_backward_rasterize[_lit]: K6's line setup, its band kernel and the K7 / K8 gather as
                                         launches of their own, one after the other -- the order every call of more than 96 k
                                         faces (batch x faces) takes anyway.  Smaller calls with texture_size <= 13 put the
                                         gather and the zeros of grad_textures into ONE launch with the line setup, in front of
                                         the band kernel (both only need the visible-face lists), and add K6's sums onto
                                         grad_faces last.  Same values: one float addition per element of grad_faces either
                                         way; a testing / measuring aid. */

#define NR_FLAG_K6_LEGACY 128          /* K6, either arithmetic mode: the piece-per-lane band kernel of rounds 3-4 (k_bpm_fast) instead of
                                         k_bpm_row.  A testing / measuring aid: same sweeps, terms rounded (default mode) and summed
                                         another way (both within the mode's bound of the oracle). */
#define NR_FLAG_K6_PX 65536              /* accepted and ignored since 0.6.0 (it forced round 5's lane-parallel kernel, whose place
                                         k_bpm_row has taken for every call: both arithmetic modes have one band kernel -- 16 lanes
                                         per line record, four records per wave instruction -- wherever its band fits the LDS
                                         (raster <= 1024; the default mode: eps > 0); the scan path, larger rasters and the default
                                         mode with eps = 0 run on k_bpm_fast).  Which kernel a call takes does not depend on its
                                         batch size. */

/*
 * faces_z_ref (nr_forward_texture_sampling, nr_forward_rasterize, nr_backward_textures, nr_backward_rasterize):
 * the reference samples textures with the vertex depths of BATCH ELEMENT 0 (`&faces[face_index * 9]`, rasterize.py:389,
 * SURVEY Q1).  NULL = batch element 0 of `faces`, i.e. the reference's behaviour for a call that holds the whole
 * batch.  A caller that holds only a SHARD of the batch (views [start, stop) on one GPU) passes the [F,3,3] faces of
 * the GLOBAL batch element 0 here (device memory; neural_renderer_amd.distributed.broadcast_reference_faces), so that
 * sharded and unsharded runs give identical bits.  Ignored when NR_FLAG_FIX_TEXTURE_BATCH_Z is set.
 *
 * visible_faces (uint8 [B,F], optional everywhere): 1 for every face that owns at least one pixel of its image.  The
 * forward writes every element when the pointer is given; the K6 pipeline starts from it (it otherwise rebuilds the flags
 * with one more pass over face_index_map).  It is a residual like face_index_map: pass back what the forward produced.
 */

int nr_version(void);
const char *nr_error_string(int code);

/* Scratch needed by the forward: the packed 64-bit z-buffer (depth bits << 32 | face index, one word per pixel) and the
 * queue of faces with large screen boxes.  The reference's `faces_inv` scratch (rasterize.py:240) is not materialised. */
size_t nr_forward_workspace_bytes(int32_t batch_size, int32_t num_faces, int32_t image_size);

/* Scratch needed by nr_backward_pixel_map / nr_backward_rasterize: per image the sorted list of visible faces, their edge
 * line ranges, face -> list position, six double sums per listed face and the band line records (+ the flags when
 * visible_faces is not passed).  With return_rgb == return_alpha == 0 (a depth-only nr_backward_rasterize: no K6) only the
 * lists are needed and the size is B * F * 4 bytes + a header. */
size_t nr_backward_workspace_bytes(int32_t batch_size, int32_t num_faces, int32_t image_size,
                                   int32_t return_rgb, int32_t return_alpha);

/*
 * Visibility (K1+K2, rasterize.py:240-359; tie rule "min depth, then lowest face index").
 * Writes EVERY element of face_index_map (-1 where empty), weight_map (0), depth_map (far) and, when
 * non-NULL, face_inv_map (0) -- the caller need not pre-fill them (the reference does, :478-496).
 * weight_map, depth_map, face_inv_map and visible_faces may each be NULL when the caller does not need them.
 * near / far may be any numbers (rasterize.py:331 pastes them as literals): depths enter the packed z-buffer through an
 * order-preserving integer key, so near <= 0 (faces behind the camera, negative depths) behaves as in the reference.
 */
int nr_forward_face_index_map(const float *faces, int32_t *face_index_map, float *weight_map, float *depth_map,
                              float *face_inv_map, uint8_t *visible_faces, int32_t batch_size, int32_t num_faces,
                              int32_t image_size, double near, double far, void *workspace, size_t workspace_bytes,
                              void *stream);

/*
 * Shading (K4 + K5, rasterize.py:361-465): trilinear sampling of the winning face's texture cube,
 * background blending and alpha.  rgb_map (needs faces, textures, weight_map, depth_map, background)
 * and alpha_map are each optional (NULL = not requested) but at least one must be given.
 * background: 3 floats, or batch_size*3 floats when bg_per_batch != 0 (device memory).
 * sampling_index_map / sampling_weight_map: optional residuals of the reference (:394-395); zero where empty.
 */
int nr_forward_texture_sampling(const float *faces, const float *faces_z_ref, const float *textures,
                                const int32_t *face_index_map, const float *weight_map, const float *depth_map,
                                float *rgb_map, int32_t *sampling_index_map, float *sampling_weight_map,
                                const float *background, int32_t bg_per_batch, float *alpha_map, int32_t batch_size,
                                int32_t num_faces, int32_t image_size, int32_t texture_size, double eps, int32_t flags,
                                void *stream);

/*
 * Approximate gradient of rgb / alpha w.r.t. vertex x, y (K6, rasterize.py:517-748).
 * STORES every element of grad_faces [B,F,3,3] (z components and back faces = 0), like the reference
 * (:736 after the zero fill of :851).  rgb_map must be the post-background map.
 * return_rgb / return_alpha select the terms; the matching map and gradient pointers must be non-NULL.
 * flags: NR_FLAG_EXACT_GRADIENT, NR_FLAG_K6_GLOBAL; visible_faces: the forward's flags or NULL.
 */
int nr_backward_pixel_map(const float *faces, const int32_t *face_index_map, const float *rgb_map,
                          const float *alpha_map, const float *grad_rgb_map, const float *grad_alpha_map,
                          float *grad_faces, int32_t batch_size, int32_t num_faces, int32_t image_size, double eps,
                          int32_t return_rgb, int32_t return_alpha, int32_t flags, const uint8_t *visible_faces,
                          void *workspace, size_t workspace_bytes, void *stream);

/*
 * Texture gradient (K7, rasterize.py:750-792): the sum of w * grad_rgb over the 8 taps of every pixel a
 * face owns.  STORES every element of grad_textures [B,F,ts,ts,ts,3] (zeros for faces that own no pixel):
 * the caller's zero fill (:853) is not needed.  faces is always required (screen boxes).  If both sampling
 * maps are given they are used as in the reference; if both are NULL the indices/weights are recomputed
 * from faces, weight_map, depth_map, eps and flags with the forward's arithmetic (saves 64 B/pixel of
 * residuals).
 */
int nr_backward_textures(const int32_t *face_index_map, const float *sampling_weight_map,
                         const int32_t *sampling_index_map, const float *faces, const float *faces_z_ref,
                         const float *weight_map, const float *depth_map, const float *grad_rgb_map, float *grad_textures,
                         int32_t batch_size, int32_t num_faces, int32_t image_size, int32_t texture_size, double eps,
                         int32_t flags, void *stream);

/*
 * Depth gradient (K8, rasterize.py:794-847): ACCUMULATES into grad_faces (run after nr_backward_pixel_map,
 * :881-883).  face_inv_map may be NULL: the per-face inverse is then recomputed from faces with the
 * forward's arithmetic (saves 36 B/pixel of residuals).
 */
int nr_backward_depth_map(const float *faces, const float *depth_map, const int32_t *face_index_map,
                          const float *face_inv_map, const float *weight_map, const float *grad_depth_map,
                          float *grad_faces, int32_t batch_size, int32_t num_faces, int32_t image_size, void *stream);

/*
 * Fused forward = Rasterize.forward_gpu (rasterize.py:467-513): nr_forward_face_index_map followed by
 * nr_forward_texture_sampling, identical results, one resolve pass (the winner is shaded while still in
 * registers).  rgb_map / alpha_map / weight_map / depth_map are each optional (NULL = not requested).
 */
int nr_forward_rasterize(const float *faces, const float *faces_z_ref, const float *textures, int32_t *face_index_map,
                         float *weight_map, float *depth_map, float *rgb_map, float *alpha_map, uint8_t *visible_faces,
                         const float *background, int32_t bg_per_batch, int32_t batch_size, int32_t num_faces,
                         int32_t image_size, int32_t texture_size, double near, double far, double eps, int32_t flags,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * Fused backward = Rasterize.backward_gpu (rasterize.py:849-889): K6, then K7, then K8, identical results to the
 * three stage calls above, sharing the per-image lists of visible faces that K6 builds (the gathers then
 * visit only the faces that own a pixel).  A NULL gradient pointer means "that output has no gradient"
 * (the reference substitutes zeros, :858-878): the corresponding terms are skipped.  STORES every element of
 * grad_faces and, when grad_rgb_map and grad_textures are given, of grad_textures.  Needs weight_map / depth_map
 * when grad_rgb_map or grad_depth_map is given, rgb_map / alpha_map for their gradients; workspace as for
 * nr_backward_pixel_map.  visible_faces (optional): the forward's per-face flags; K6 builds its lists from them, and with
 * only grad_depth_map given (no K6, no lists) the depth gather skips the faces they mark as owning no pixel.
 */
int nr_backward_rasterize(const float *faces, const float *faces_z_ref, const int32_t *face_index_map,
                          const float *weight_map, const float *depth_map, const float *rgb_map, const float *alpha_map,
                          const float *grad_rgb_map, const float *grad_alpha_map, const float *grad_depth_map,
                          float *grad_faces, float *grad_textures, int32_t batch_size, int32_t num_faces,
                          int32_t image_size, int32_t texture_size, double eps, int32_t flags,
                          const uint8_t *visible_faces, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Per-face light colours instead of lit, duplicated textures.  Renderer.render (reference renderer.py:77-103)
 * hands the rasterizer textures that fill_back duplicated (:79, the copy with its first and third cube axis exchanged) and
 * lighting multiplied by one colour per face (lighting.py:50-51): 2 x B x Nf x ts^3 x 3 floats that are written, read back
 * by the few faces that own a pixel, and whose equally large gradient travels the other way.  The _lit entry points take
 * the ORIGINAL cubes plus the per-face colours and do both inside the shading:
 *     rgb(pixel of face f) = light[b, f, :] * sum_taps w * cube(f)[tap]          cube(f) = textures[b, f] for f < Nf,
 *                                                                               textures[b, f - Nf] transposed for f >= Nf
 * which is the reference's value up to the rounding order of the light product (the reference rounds light * texel per
 * texel and sums, this rounds the sum and multiplies: 1 ulp class, <= 2e-7 relative; tests/test_face_light_gpu.py).
 * Every other output is bit-identical (the geometry does not depend on textures).
 *   light [B, F, 3]; texture_faces = Nf with F == Nf or F == 2 * Nf; textures / grad_textures [B, Nf, ts,ts,ts, 3].
 *   backward: grad_textures is the gradient w.r.t. the ORIGINAL cubes (light factor included); grad_light [B, F, 3]
 *   (optional) the gradient of the colours, for nr_frontend_backward_light.  lit == NULL: exactly nr_forward_rasterize /
 *   nr_backward_rasterize.
 */
typedef struct nr_face_light {
    const float *light;     /* DEVICE [B, F, 3] */
    int32_t texture_faces;  /* Nf */
    const float *textures;  /* backward: the cubes again (DEVICE), needed when grad_light is given */
    float *grad_light;      /* backward: DEVICE [B, F, 3] or NULL; every element stored */
} nr_face_light;

int nr_forward_rasterize_lit(const nr_face_light *lit, const float *faces, const float *faces_z_ref, const float *textures,
                             int32_t *face_index_map, float *weight_map, float *depth_map, float *rgb_map, float *alpha_map,
                             uint8_t *visible_faces, const float *background, int32_t bg_per_batch, int32_t batch_size,
                             int32_t num_faces, int32_t image_size, int32_t texture_size, double near, double far,
                             double eps, int32_t flags, void *workspace, size_t workspace_bytes, void *stream);
int nr_backward_rasterize_lit(const nr_face_light *lit, const float *faces, const float *faces_z_ref,
                              const int32_t *face_index_map, const float *weight_map, const float *depth_map,
                              const float *rgb_map, const float *alpha_map, const float *grad_rgb_map,
                              const float *grad_alpha_map, const float *grad_depth_map, float *grad_faces,
                              float *grad_textures, int32_t batch_size, int32_t num_faces, int32_t image_size,
                              int32_t texture_size, double eps, int32_t flags, const uint8_t *visible_faces,
                              void *workspace, size_t workspace_bytes, void *stream);

/*
 * vertices_to_faces (reference neural_renderer/vertices_to_faces.py:4-21): faces_out[b,f,k,:] = vertices[b, faces_idx[.,f,k], :]
 * and its backward (Chainer's get_item backward = scatter-add): grad_vertices[b, faces_idx[.,f,k], :] += grad_faces[b,f,k,:]
 * with hardware float atomics; grad_vertices [B,Nv,3] is zero-filled by the call.  faces_idx is [B,Nf,3] int32 when
 * idx_per_batch != 0, else one [Nf,3] topology shared by the batch.  Indices must lie in [0, Nv): the Python binding checks
 * that (IndexError, like the reference's get_item); the kernels clamp, so a bad index never leaves the buffers.
 */
int nr_vertices_to_faces(const float *vertices, const int32_t *faces_idx, float *faces_out, int32_t batch_size,
                         int32_t num_vertices, int32_t num_faces, int32_t idx_per_batch, void *stream);
int nr_vertices_to_faces_backward(const float *grad_faces, const int32_t *faces_idx, float *grad_vertices,
                                  int32_t batch_size, int32_t num_vertices, int32_t num_faces, int32_t idx_per_batch,
                                  void *stream);

/*
 * Image epilogue of rasterize_rgbad (reference rasterize.py:953-969), one bandwidth-bound kernel per direction:
 * rgb [B,S,S,3] -> [B,3,is,is] (NHWC -> NCHW), alpha / depth [B,S,S] -> [B,is,is], every output flipped vertically
 * (row 0 of the maps is the bottom row, row 0 of the images the top row) and, when anti_aliasing != 0, averaged over
 * 2x2 blocks (is = S/2, S even; else is = S).  Each map / image pair is optional (both NULL = not requested).
 * The backward writes every element of the requested map gradients.
 */
int nr_image_epilogue(const float *rgb_map, const float *alpha_map, const float *depth_map, float *rgb_out,
                      float *alpha_out, float *depth_out, int32_t batch_size, int32_t image_size, int32_t anti_aliasing,
                      void *stream);
int nr_image_epilogue_backward(const float *grad_rgb_out, const float *grad_alpha_out, const float *grad_depth_out,
                               float *grad_rgb_map, float *grad_alpha_map, float *grad_depth_map, int32_t batch_size,
                               int32_t image_size, int32_t anti_aliasing, void *stream);

/*
 * Fused geometry + lighting front-end of Renderer.render / render_silhouettes / render_depth
 * (reference renderer.py:35-107): fill_back (:37-38, :77-79), lighting (lighting.py:8-51), look_at (look_at.py:7-46) or
 * look (look.py:7-45), perspective (perspective.py:5-19) and vertices_to_faces (vertices_to_faces.py:4-21) in one kernel,
 * and their whole backward (including the light -> normal -> vertex path, the face -> vertex scatter and the gradient of
 * the camera position) in one kernel plus a per-image camera kernel.
 *
 *   vertices [B,Nv,3] world space; faces_idx int32 [B,Nf,3] (idx_per_batch != 0) or [Nf,3]; textures [B,Nf,ts,ts,ts,3] or
 *   NULL (silhouette / depth rendering: no lighting); eye: DEVICE pointer, [B,3] (eye_per_batch != 0) or [3];
 *   faces_out [B,F,3,3] with F = Nf * (fill_back ? 2 : 1): face f and, at Nf + f, its copy with reversed vertex order;
 *   textures_out [B,F,ts,ts,ts,3]: textures * light and, at Nf + f, the (i,j,k) -> (k,j,i) transposed textures * the
 *   light of the reversed face.  nr_camera / nr_light live in HOST memory and are read during the call.
 */
#define NR_CAMERA_LOOK_AT 1 /* rotation from eye -> target, target = `at` (look_at.py) */
#define NR_CAMERA_LOOK 2    /* rotation from a fixed viewing direction, target = `direction` (look.py) */

typedef struct nr_camera {
    int32_t mode;        /* NR_CAMERA_LOOK_AT or NR_CAMERA_LOOK */
    int32_t perspective; /* != 0: x/z/width, y/z/width (perspective.py:15-17) */
    float target[3];     /* `at` or `direction` */
    float up[3];
    float width;         /* tan(viewing_angle / 180 * 3.1416), computed by the host in float32 (perspective.py:10-13) */
} nr_camera;

typedef struct nr_light { /* lighting.py:9-13 */
    float intensity_ambient, intensity_directional;
    float color_ambient[3], color_directional[3], direction[3];
} nr_light;

/* Scratch for nr_frontend_backward when grad_eye is requested (per-image camera sums). */
size_t nr_frontend_workspace_bytes(int32_t batch_size);

int nr_frontend_forward(const float *vertices, const int32_t *faces_idx, const float *textures, const float *eye,
                        float *faces_out, float *textures_out, int32_t batch_size, int32_t num_vertices,
                        int32_t num_faces, int32_t texture_size, int32_t idx_per_batch, int32_t eye_per_batch,
                        int32_t fill_back, const nr_camera *camera, const nr_light *light, void *stream);

/*
 * grad_faces [B,F,3,3] and grad_textures_out [B,F,ts,ts,ts,3] (NULL = the lit textures received no gradient) are the
 * gradients of the two outputs.  Each result is optional (NULL = not needed): grad_vertices [B,Nv,3] (zero-filled by the
 * call, accumulated with hardware float atomics), grad_textures [B,Nf,ts,ts,ts,3] (every element stored; needs
 * grad_textures_out), grad_eye [B,3] or [3] (needs grad_vertices and the workspace).  `at`, `up`, `direction`, the
 * viewing angle and the light parameters are constants of the call (no gradients), as in Renderer.
 */
int nr_frontend_backward(const float *vertices, const int32_t *faces_idx, const float *textures, const float *eye,
                         const float *grad_faces, const float *grad_textures_out, float *grad_vertices,
                         float *grad_textures, float *grad_eye, int32_t batch_size, int32_t num_vertices,
                         int32_t num_faces, int32_t texture_size, int32_t idx_per_batch, int32_t eye_per_batch,
                         int32_t fill_back, const nr_camera *camera, const nr_light *light, void *workspace,
                         size_t workspace_bytes, void *stream);

/*
 * The same front-end for the _lit rasterizer entry points: no textures in or out.  light_out [B,F,3] receives the colour of
 * every face (lighting.py:28-47) and of its reversed copy; nr_frontend_backward_light takes the gradient of those colours
 * (grad_light [B,F,3], e.g. from nr_backward_rasterize_lit; NULL = none) in place of grad_textures_out.  The gradient of the
 * textures does not pass through here (nr_backward_rasterize_lit stores it).
 */
int nr_frontend_forward_light(const float *vertices, const int32_t *faces_idx, const float *eye, float *faces_out,
                              float *light_out, int32_t batch_size, int32_t num_vertices, int32_t num_faces,
                              int32_t idx_per_batch, int32_t eye_per_batch, int32_t fill_back, const nr_camera *camera,
                              const nr_light *light, void *stream);
int nr_frontend_backward_light(const float *vertices, const int32_t *faces_idx, const float *eye, const float *grad_faces,
                               const float *grad_light, float *grad_vertices, float *grad_eye, int32_t batch_size,
                               int32_t num_vertices, int32_t num_faces, int32_t idx_per_batch, int32_t eye_per_batch,
                               int32_t fill_back, const nr_camera *camera, const nr_light *light, void *workspace,
                               size_t workspace_bytes, void *stream);

/*
 * Texture baking of load_obj(load_texture=True) (K10, reference load_obj.py:87-144): for every texel (i0,i1,i2) of every
 * face with is_update[f] != 0, textures[f,i0,i1,i2,:] = bilinear lookup of `image` at the barycentric point
 * (i0,i1,i2)/(i0+i1+i2) of the face's uv triangle.  image [H,W,3] float32 in [0,1], ALREADY flipped vertically (:85);
 * faces_uv [Nf,3,2]; textures [Nf,ts,ts,ts,3] in/out (other faces untouched).  Texel (0,0,0) of an updated face is NaN, as
 * in the reference (0/0); reads outside the image are clamped to the nearest pixel (zero weight when the reference is defined).
 */
int nr_load_textures(const float *image, const float *faces_uv, const int32_t *is_update, float *textures,
                     int32_t num_faces, int32_t texture_size, int32_t image_height, int32_t image_width, void *stream);

/*
 * Texture atlas of save_obj(..., textures) (K11, reference save_obj.py:10-146): image [tile_height*tso, tile_width*tso, 3]
 * (NOT yet flipped) from textures [Nf,tsi,tsi,tsi,3] and the per-face tile triangles tile_vertices [Nf,3,2] in atlas pixel
 * coordinates (save_obj.py:17-25); tiles beyond the last face are written as 0.  Includes the seam pass (:115-146).
 */
int nr_create_texture_image(const float *textures, const float *tile_vertices, float *image, int32_t num_faces,
                            int32_t texture_size_in, int32_t texture_size_out, int32_t tile_width, int32_t tile_height,
                            void *stream);

/*
 * Masked Adam update (reference optimizers.py:17-34): for every element with grad != 0
 *   m += one_minus_beta1 * (grad - m);  v += one_minus_beta2 * (grad * grad - v);  v = max(v, 0);
 *   param -= lr * m / (sqrt(v) + eps);
 * elements with a zero gradient keep parameter and moments.  float32, in place; `lr` = alpha_t * the parameter's multiplier.
 */
int nr_adam_update(float *param, const float *grad, float *m, float *v, size_t count, float lr, float one_minus_beta1,
                   float one_minus_beta2, float eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NR_HIP_H */
