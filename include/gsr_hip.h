/*
 * gsr_hip.h — C ABI of libgsr_hip.so, the MI355X (gfx950) differentiable Gaussian rasterizer.
 *
 * This is the drop-in boundary below the reference's PyTorch extension modules.  Every entry
 * point takes plain device pointers, sizes and a hipStream_t (passed as void*); there are no
 * torch types.  All device pointers must be valid on the current HIP device; "may be NULL"
 * is stated per argument.  All arithmetic is fp32; integer outputs are int32/uint32/uint8.
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   DSR = submodules/diff-surfel-rasterization, DGR = submodules/diff-gaussian-rasterization,
 *   CME = submodules/cubemapencoder.
 *
 * Return convention: >= 0 success (forward entry points return num_rendered), < 0 error code
 * (GSR_E_*); gsr_last_error() returns a thread-local description of the last failure.
 *
 * Workspace protocol: the three opaque per-call buffers the reference grows through
 * std::function<char*(size_t)> callbacks (DSR rasterize_points.cu:31-37, rasterizer.h:27-29)
 * are requested through `gsr_alloc_fn`: it is called at most once per `which` per forward
 * call with the exact byte size and must return a device pointer aligned to >= 256 bytes that
 * stays valid until the matching backward call has completed.  The layout inside the buffers
 * is private to this library and differs from the reference's.
 *
 * Alignment: tensors that are accessed with 128-bit loads/stores must be 16-byte aligned — shs
 * and dL_dsh when a row (3*M floats) is a multiple of 16 bytes (M = 4, 8, 12, 16), dL_drot,
 * and variant G's dL_dconic.  Fresh torch allocations always are; VIEWS into a packed buffer
 * (gradient sinks) are only if every slice starts at a multiple of 4 floats.  A misaligned
 * pointer is refused with GSR_E_INVALID (no kernel is launched).
 */
#ifndef GSR_HIP_H_
#define GSR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_BUF_GEOM 0
#define GSR_BUF_BINNING 1
#define GSR_BUF_IMAGE 2

#define GSR_E_INVALID (-1)     /* bad argument (NULL where required, negative size) */
#define GSR_E_HIP (-2)         /* a HIP runtime call or kernel launch failed */
#define GSR_E_ALLOC (-3)       /* gsr_alloc_fn returned NULL */
#define GSR_E_PREFILTERED (-4) /* a Gaussian was culled although prefiltered was set (reference: __trap()) */
#define GSR_E_NONRGB (-5)      /* NUM_CHANNELS != 3 without precomputed colours (never: library is built for 3) */

typedef void* (*gsr_alloc_fn)(void* user, int which, size_t bytes);

const char* gsr_last_error(void);
/* ABI version of the library = GSR_ABI_VERSION of the header it was built from.  A binding checks gsr_version() == GSR_ABI_VERSION at load
 * (gaussian-splatting-reflection_amd/_gsr.py does) so that a caller compiled against another header fails loudly instead of passing shifted
 * arguments.  Rule: an exported entry point NEVER changes its signature; extensions get new symbols (…_accum, …_ex, …_keys, …_refl).
 *   100  rounds 1-3.  Round 3 broke the rule once: gsr_deferred_reflection_forward_ex / _backward_ex gained a sort_keys argument in place.
 *   101  round 4: those two are back to their round-2 signatures (no sort_keys) and the forms with keys are the new symbols
 *        gsr_deferred_reflection_forward_keys / _backward_keys; new: gsr_gauss_backward_accum, gsr_surfel_forward_refl. */
#define GSR_ABI_VERSION 101
int gsr_version(void);

/* ---------------------------------------------------------------------------------------------
 * Variant S — 2D Gaussian surfels.  Replaces CudaRasterizer::Rasterizer::forward
 * (DSR cuda_rasterizer/rasterizer.h:24-56, rasterizer_impl.cu:198-355).
 *   env_scope_mask  uint8[P] (bool) or NULL (= all false)
 *   shs             float[P,M,3] or NULL when colors_precomp is given
 *   scales float[P,2], rotations float[P,4] or NULL when transMat_precomp float[P,9] is given
 *   out_color float[3,H,W]; out_others float[8,H,W]; out_refl_strength_map float[H,W];
 *   radii int32[P]; gaussian_weights float[P].  All outputs are fully written.
 * Returns num_rendered. */
int gsr_surfel_forward(gsr_alloc_fn alloc, void* alloc_user, int P, int D, int M, const float* background, int width,
                       int height, const float* means3D, const uint8_t* env_scope_mask, const float* shs,
                       const float* colors_precomp, const float* refl_strengths, const float* opacities,
                       const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                       float tan_fovy, int prefiltered, float* out_color, float* out_others,
                       float* out_refl_strength_map, int* radii, float* gaussian_weights, int debug, void* stream);

/* Replaces CudaRasterizer::Rasterizer::backward (DSR rasterizer.h:58-95, rasterizer_impl.cu:358-466).
 *   R = num_rendered returned by the forward call; geom/binning/image = the forward's buffers.
 *   dL_dpix float[3,H,W]; dL_dothers float[8,H,W] (planes 0-6 read); dL_drefl_strength_map float[H,W]
 *   Outputs (fully written, no pre-zeroing needed): dL_dmean2D[P,3] dL_dnormal[P,3] dL_dopacity[P]
 *   dL_dcolor[P,3] dL_drefl_strengths[P] dL_dmean3D[P,3] dL_dtransMat[P,9] dL_dsh[P,M,3]
 *   dL_dscale[P,2] dL_drot[P,4].
 *   Extension: dL_dnormal may be NULL, dL_dcolor may be NULL when shs is given, dL_dtransMat may be NULL when scales / rotations
 *   are given: those per-view gradients then belong to inputs the caller did not supply and are not written (the reference fills
 *   zero-initialised tensors for them, rasterize_points.cu:207-226, which its autograd wrapper then drops). */
int gsr_surfel_backward(int P, int D, int M, int R, const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp,
                        const float* refl_strengths, const float* scales, float scale_modifier, const float* rotations,
                        const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer,
                        void* binning_buffer, void* image_buffer, const float* dL_dpix, const float* dL_dothers,
                        const float* dL_drefl_strength_map, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
                        float* dL_dcolor, float* dL_drefl_strengths, float* dL_dmean3D, float* dL_dtransMat,
                        float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream);
/* Extension (no reference counterpart; the reference trains one view per step, train.py:134-150): the same call with one
 * more switch.  accumulate != 0: the six PARAMETER gradients dL_dopacity, dL_drefl_strengths, dL_dmean3D, dL_dsh,
 * dL_dscale, dL_drot are ADDED to the given tensors by the per-Gaussian kernel (stream-ordered read-add-write, no atomics)
 * instead of written, so that several views of a batch accumulate into one gradient buffer on the device before ONE
 * all-reduce (BASELINE config 4: 8 views over N GPUs).  The per-view outputs dL_dmean2D, dL_dnormal, dL_dcolor,
 * dL_dtransMat are overwritten in both modes. */
int gsr_surfel_backward_accum(int P, int D, int M, int R, const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp,
                        const float* refl_strengths, const float* scales, float scale_modifier, const float* rotations,
                        const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer,
                        void* binning_buffer, void* image_buffer, const float* dL_dpix, const float* dL_dothers,
                        const float* dL_drefl_strength_map, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
                        float* dL_dcolor, float* dL_drefl_strengths, float* dL_dmean3D, float* dL_dtransMat,
                        float* dL_dsh, float* dL_dscale, float* dL_drot, int accumulate, int debug, void* stream);
/* Extension: one more upstream gradient.  dL_dnormal_extra float[3,H,W] or NULL is a SECOND gradient of the blended
 * normal planes (planes 2..4 of out_others) — the one the deferred-reflection pass returns for its normal_view input
 * (gaussian_renderer/__init__.py:25-35 of the reference reads allmap[2:5]) — and the tile kernel adds it to planes 2..4 of
 * dL_dothers as it loads them.  Plain autograd sums the two with a zero-fill, a slice copy and an add over the
 * 8-plane image (0.05 ms per 1080p view); here that pass does not exist.  NULL: gsr_surfel_backward_accum. */
int gsr_surfel_backward_ex(int P, int D, int M, int R, const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp,
                        const float* refl_strengths, const float* scales, float scale_modifier, const float* rotations,
                        const float* transMat_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer,
                        void* binning_buffer, void* image_buffer, const float* dL_dpix, const float* dL_dothers,
                        const float* dL_drefl_strength_map, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
                        float* dL_dcolor, float* dL_drefl_strengths, float* dL_dmean3D, float* dL_dtransMat,
                        float* dL_dsh, float* dL_dscale, float* dL_drot, int accumulate,
                        const float* dL_dnormal_extra, int debug, void* stream);

/* Extension (round 4): the forward of rasterizer + deferred reflection in ONE pass over the pixels.  The reference runs the rasterizer,
 * then ~12 torch ops per pixel (gaussian_renderer/__init__.py:22-35,143-199); gsr_deferred_reflection_forward below fuses those ops into
 * one kernel; here that kernel's per-pixel code runs INSIDE the rasterizer's forward tile kernel, as its epilogue (a wave that has
 * finished its 8x8 pixel block has normal, base colour and reflection strength in registers), and the texel-interleaved cubemap copy is
 * made by the per-Gaussian kernel: no pixel kernel, no interleave dispatch, no re-read of seven planes.  All outputs of
 * gsr_surfel_forward are still written.  The backward stays two calls (gsr_deferred_reflection_backward_keys, then
 * gsr_surfel_backward_ex): its texel-gradient tail needs every pixel's record EARLY, so that it can hide beside the tile backward — run
 * as a prologue of that kernel (built and measured, round 4) the records are complete only when the kernel ends and the tail is exposed.
 *   refl == NULL: gsr_surfel_forward exactly.
 * Forward descriptor: cam, cubemap [6,3,L,L], fail_value [3], L as gsr_deferred_reflection_forward; cubemap_rgba: 6*L*L*4 floats, 16-byte
 * aligned, FILLED by the call (texel-interleaved copy, made by the per-Gaussian kernel) and to be handed to the backward; out_final,
 * out_refl_color, out_normal_world [3,H,W]; sort_keys: NULL or width*height uint32 (see gsr_deferred_reflection_forward_keys).
 * scratch (optional, with sort_keys): the buffer gsr_deferred_reflection_backward_keys will get.  The forward then also sorts the
 * (key, pixel) pairs into it — with async_sort on the side stream, where the sort runs beside whatever follows the forward on `stream`
 * (the loss; the start of the backward) while CUs are still free: 1024-thread sort workgroups cannot start on a chip the tile backward
 * has filled.  Hand the SAME scratch, untouched, to the backward with keys_sorted = 1; it must stay alive until gsr_side_join. */
typedef struct {
	const float* cam;
	const float* cubemap;
	const float* fail_value;
	uint32_t L;
	float* cubemap_rgba;
	float* out_final;
	float* out_refl_color;
	float* out_normal_world;
	uint32_t* sort_keys;
	float* scratch;          /* NULL, or (with sort_keys) the scratch of the backward: the call also SORTS the keys into it */
	size_t scratch_floats;   /* gsr_deferred_reflection_scratch_floats(L, W, H, 1) */
	int async_sort;          /* != 0: that sort runs on the library's side stream, forked behind the tile kernel */
} gsr_refl_forward;
int gsr_surfel_forward_refl(gsr_alloc_fn alloc, void* alloc_user, int P, int D, int M, const float* background, int width,
                       int height, const float* means3D, const uint8_t* env_scope_mask, const float* shs,
                       const float* colors_precomp, const float* refl_strengths, const float* opacities,
                       const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                       float tan_fovy, int prefiltered, float* out_color, float* out_others,
                       float* out_refl_strength_map, int* radii, float* gaussian_weights, const gsr_refl_forward* refl,
                       int debug, void* stream);
/* ---------------------------------------------------------------------------------------------
 * Variant G — 3D Gaussians with EWA projection, anti-aliasing and inverse depth.  Replaces
 * CudaRasterizer::Rasterizer::forward (DGR cuda_rasterizer/rasterizer.h:24-57, rasterizer_impl.cu:198-349).
 *   normals float[P,3], refl_strengths float[P] are required.
 *   out_color[3,H,W] out_normal_map[3,H,W] out_refl_strength_map[H,W] out_invdepth[H,W] or NULL, radii int32[P]. */
int gsr_gauss_forward(gsr_alloc_fn alloc, void* alloc_user, int P, int D, int M, const float* background, int width,
                      int height, const float* means3D, const float* shs, const float* colors_precomp,
                      const float* normals, const float* refl_strengths, const float* opacities, const float* scales,
                      float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                      const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                      float* out_color, float* out_normal_map, float* out_refl_strength_map, float* out_invdepth,
                      int antialiasing, int* radii, int debug, void* stream);

/* Replaces CudaRasterizer::Rasterizer::backward (DGR rasterizer.h:59-101, rasterizer_impl.cu:353-472).
 *   dL_invdepths may be NULL (then dL_dinvdepth is not touched).  dL_dmean2D_pixels is what the
 *   reference hands to Python as grad_means2D (DGR rasterize_points.cu:263).  dL_dconic is float[P,4]
 *   (slots x,y,w used).  Outputs are fully written.  *   Extension: dL_dmean2D and dL_dconic (intermediates of the reference's backward that its binding never returns) may be NULL, dL_dcolor
 *   may be NULL when shs is given and dL_dcov3D when scales / rotations are given: not written then. */
int gsr_gauss_backward(int P, int D, int M, int R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* normals,
                       const float* refl_strengths, const float* opacities, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                       void* geom_buffer, void* binning_buffer, void* image_buffer, const float* dL_dpix,
                       const float* dL_dnormal_map, const float* dL_drefl_strength_map, const float* dL_invdepths,
                       float* dL_dmean2D, float* dL_dmean2D_pixels, float* dL_dconic, float* dL_dopacity,
                       float* dL_dcolor, float* dL_dnormals, float* dL_drefl_strengths, float* dL_dinvdepth,
                       float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                       int antialiasing, int debug, void* stream);
/* Extension (round 4; the counterpart of gsr_surfel_backward_accum): accumulate != 0 ADDS the seven PARAMETER gradients dL_dopacity,
 * dL_dnormals, dL_drefl_strengths, dL_dmean3D, dL_dsh, dL_dscale, dL_drot to the given tensors (stream-ordered read-add-write in the
 * per-Gaussian kernel, no atomics) instead of writing them: several views of a batch accumulate into one gradient buffer on the device.
 * The per-view outputs dL_dmean2D, dL_dmean2D_pixels, dL_dconic, dL_dcolor, dL_dinvdepth, dL_dcov3D are overwritten in both modes. */
int gsr_gauss_backward_accum(int P, int D, int M, int R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* normals,
                       const float* refl_strengths, const float* opacities, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                       void* geom_buffer, void* binning_buffer, void* image_buffer, const float* dL_dpix,
                       const float* dL_dnormal_map, const float* dL_drefl_strength_map, const float* dL_invdepths,
                       float* dL_dmean2D, float* dL_dmean2D_pixels, float* dL_dconic, float* dL_dopacity,
                       float* dL_dcolor, float* dL_dnormals, float* dL_drefl_strengths, float* dL_dinvdepth,
                       float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                       int antialiasing, int accumulate, int debug, void* stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible (DSR rasterizer.h:19-23, rasterizer_impl.cu:141-153). */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * Introspection of the private workspace (used by the parity tests; a maintainer never needs it).
 * Copies the named per-Gaussian / per-instance / per-pixel array from the workspaces of a
 * completed forward call into `dst` (device pointer).  variant: 0 = S, 1 = G.
 * Names: "depths" f32[P], "means2D" f32[P,2], "tiles_touched" u32[P], "point_offsets" u32[P],
 * "clamped" u8[P,3], "rgb" f32[P,3], "geom4" f32[P,4] (conic_opacity / normal_opacity),
 * "transMat" f32[P,9] (S), "cov3D" f32[P,6] (G), "point_list" u32[R], "keys" u64[R],
 * "ranges" u32[tiles,2], "final_T" f32[planes,H,W], "n_contrib" u32[planes,H,W]. */
int gsr_debug_fetch(int variant, const char* name, int P, int R, int width, int height, const void* geom_buffer,
                    const void* binning_buffer, const void* image_buffer, void* dst, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Cubemap encoder.  Replaces cubemap_encode_forward / cubemap_encode_backward
 * (CME src/cubemapencoder.h:6-17, cubemapencoder.cu:430-488, 713-779).  fp32 only.
 *   inputs float[B,3]; cubemap float[6,C,L,L]; fail_value float[C]; outputs float[C,B] (channel-major).
 *   interp: 0 nearest, 1 bilinear; seamless: 0/1. */
int gsr_cubemap_forward(const float* inputs, const float* cubemap, const float* fail_value, float* outputs,
                        uint32_t interp, uint32_t seamless, uint32_t B, uint32_t C, uint32_t L, void* stream);
/*   grad_cubemap [6,C,L,L] and grad_fail [C] are accumulated into (the caller zeroes them, as
 *   cubemap_encoder.py:53-55 does); grad_inputs [B,3] is fully written. */
int gsr_cubemap_backward(const float* grad_outputs, const float* inputs, const float* cubemap, float* grad_cubemap,
                         float* grad_inputs, float* grad_fail, uint32_t interp, uint32_t seamless, uint32_t B,
                         uint32_t C, uint32_t L, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused deferred-reflection pixel pass.  Replaces the ~12 torch ops of
 * gaussian_renderer/__init__.py:22-35,148,178-179,197-199 + utils/general_utils.py:177-197:
 *   n_world = normalize(N_view . W2V[:3,:3]^T) (+1e-6), d = normalize((K^-1 [x,y,1] - T) . R^T ... ),
 *   r = d - 2 n (d.n), c = sigmoid(cubemap(r)), final = (1-s) base + s c.
 *   normal_view float[3,H,W] (allmap planes 2..4); base_color float[3,H,W]; refl_strength float[H,W];
 *   cam: float[25] = viewmatrix 3x3 block (row-major, 9) | Kinv (9, row-major) | R_w2c (9, row-major) | T (3) |
 *   campos filled by the host (see host code); outputs final float[3,H,W], refl_color float[3,H,W],
 *   normal_world float[3,H,W] (normalised). C must be 3. */
int gsr_deferred_reflection_forward(const float* normal_view, const float* base_color, const float* refl_strength,
                                    const float* cam, const float* cubemap, const float* fail_value, uint32_t L,
                                    int width, int height, float* out_final, float* out_refl_color,
                                    float* out_normal_world, void* stream);
/*   Upstream grads: g_final [3,H,W] (required), g_refl_color, g_normal_world may be NULL.
 *   Outputs: g_normal_view [3,H,W], g_base [3,H,W], g_strength [H,W], g_cubemap [6,3,L,L], g_fail [3]: all fully
 *   written.  scratch: caller-provided device buffer of `scratch_floats` floats (contents ignored).
 *   With at least gsr_deferred_reflection_scratch_floats(L, W, H, 0) = (6*L*L+1)*4 floats the texel gradients are added
 *   with float atomics from the pixel kernel; with gsr_deferred_reflection_scratch_floats(L, W, H, 1) floats
 *   (~44 bytes per pixel + the sort's temporary storage; 16-byte aligned) the sorted path runs instead: one footprint
 *   record per pixel, a radix sort by texel id, runs of equal texels summed in registers and LDS. */
size_t gsr_deferred_reflection_scratch_floats(uint32_t L, int width, int height, int binned);
int gsr_deferred_reflection_backward(const float* normal_view, const float* base_color, const float* refl_strength,
                                     const float* cam, const float* cubemap, const float* fail_value, uint32_t L,
                                     int width, int height, const float* g_final, const float* g_refl_color,
                                     const float* g_normal_world, float* g_normal_view, float* g_base, float* g_strength,
                                     float* g_cubemap, float* g_fail, float* scratch, size_t scratch_floats, void* stream);
/* Extension: accumulate != 0 adds the cubemap / fail-value gradient to g_cubemap / g_fail instead of writing them
 * (several views per optimizer step, see gsr_surfel_backward_accum). */
int gsr_deferred_reflection_backward_accum(const float* normal_view, const float* base_color, const float* refl_strength,
                                     const float* cam, const float* cubemap, const float* fail_value, uint32_t L,
                                     int width, int height, const float* g_final, const float* g_refl_color,
                                     const float* g_normal_world, float* g_normal_view, float* g_base, float* g_strength,
                                     float* g_cubemap, float* g_fail, float* scratch, size_t scratch_floats,
                                     int accumulate, void* stream);
/* Extension: async_tail != 0 (sorted path only) enqueues the part of the backward that produces g_cubemap / g_fail — sort of
 * the footprint records, run combine, unpack; nothing else in a training step depends on it before the optimizer — on a
 * side stream owned by the library (one per device, highest priority) that forks from `stream` after the pixel kernel, so that
 * it runs beside whatever the caller enqueues next on `stream` (the rasterizer backward).  g_normal_view / g_base /
 * g_strength are complete in `stream` order as always.  g_cubemap, g_fail and scratch must stay untouched and alive until
 * gsr_side_join(s) has been called: it makes stream s wait (device-side, no host block) for all side work enqueued so far
 * on the current device.  Successive tails are ordered among themselves (accumulate over a batch of views works).
 * cubemap_rgba: NULL, or the texel-interleaved copy gsr_deferred_reflection_forward_ex made of the SAME cubemap. */
int gsr_deferred_reflection_backward_ex(const float* normal_view, const float* base_color, const float* refl_strength,
                                        const float* cam, const float* cubemap, const float* fail_value, uint32_t L,
                                        int width, int height, const float* g_final, const float* g_refl_color,
                                        const float* g_normal_world, float* g_normal_view, float* g_base, float* g_strength,
                                        float* g_cubemap, float* g_fail, float* scratch, size_t scratch_floats,
                                        int accumulate, int async_tail, const float* cubemap_rgba, void* stream);
/* ... and with sort_keys (NULL, or the width*height keys gsr_deferred_reflection_forward_keys / gsr_surfel_forward_refl wrote for the SAME
 * inputs): the sort of the footprint records then depends on nothing this call computes; with async_tail it forks before the pixel kernel
 * and runs beside it, and only the run combine waits for the records.  The keys must stay alive and untouched like scratch.
 * keys_sorted != 0: gsr_surfel_forward_refl has already sorted the keys into THIS scratch (its `scratch` field): no sort here. */
int gsr_deferred_reflection_backward_keys(const float* normal_view, const float* base_color, const float* refl_strength,
                                        const float* cam, const float* cubemap, const float* fail_value, uint32_t L,
                                        int width, int height, const float* g_final, const float* g_refl_color,
                                        const float* g_normal_world, float* g_normal_view, float* g_base, float* g_strength,
                                        float* g_cubemap, float* g_fail, float* scratch, size_t scratch_floats,
                                        int accumulate, int async_tail, const float* cubemap_rgba, const uint32_t* sort_keys,
                                        int keys_sorted, void* stream);
int gsr_side_join(void* stream);
/* Extension: cubemap_rgba (NULL, or 6*L*L*4 floats, 16-byte aligned) receives a texel-interleaved copy [6][L][L][r,g,b,0] of
 * the cubemap, made by the call, from which the pixel kernel gathers each bilinear corner with one 16-byte load instead of
 * three 4-byte ones; hand the same buffer to gsr_deferred_reflection_backward_ex (cubemap_rgba) of the same cubemap, or NULL. */
int gsr_deferred_reflection_forward_ex(const float* normal_view, const float* base_color, const float* refl_strength,
                                       const float* cam, const float* cubemap, const float* fail_value, uint32_t L, int width,
                                       int height, float* out_final, float* out_refl_color, float* out_normal_world,
                                       float* cubemap_rgba, void* stream);
/* ... and with sort_keys (NULL, or width*height uint32): receives, per pixel, the sort key of the record the sorted-footprint backward
 * will make for it — the texel id of the upper-left corner of its bilinear footprint, or 6*L*L when the footprint leaves its cube
 * face or the reflection vector is zero; it depends on forward data only (see gsr_deferred_reflection_backward_keys). */
int gsr_deferred_reflection_forward_keys(const float* normal_view, const float* base_color, const float* refl_strength,
                                       const float* cam, const float* cubemap, const float* fail_value, uint32_t L, int width,
                                       int height, float* out_final, float* out_refl_color, float* out_normal_world,
                                       float* cubemap_rgba, uint32_t* sort_keys, void* stream);
/* Shading normal alone: out = normalize(normal_view rotated to world space) with the reference's +1e-6
 * (gaussian_renderer/__init__.py:148,178-179), for the initial stage where render() skips the reflection chain but still
 * returns rend_normal; `cam` as above (only its first nine floats are read).  The backward writes g_normal_view fully. */
int gsr_normal_world_forward(const float* normal_view, const float* cam, int width, int height, float* out_normal_world,
                             void* stream);
int gsr_normal_world_backward(const float* normal_view, const float* cam, int width, int height,
                              const float* g_normal_world, float* g_normal_view, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training-step passes around the rasterizer.
 *
 * gsr_ssim_l1_forward / _backward: the photometric loss of the reference's train loop, train.py:167-173
 *   loss = (1 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1 - ssim(image, gt))
 * with l1_loss = mean |x - y| (utils/loss_utils.py:40-41) and ssim = mean of the 11x11 Gaussian-window (sigma 1.5,
 * zero padding 5, per channel) SSIM map (utils/loss_utils.py:62-92).  They also stand in for the optional
 * fusedssim / fusedssim_backward pair the reference tries to import (utils/loss_utils.py:16-38).
 *   img1 (rendered), img2 (ground truth): float[C,H,W].
 *   forward: sums float[2] <- { sum |x - y|, sum ssim_map }; scratch float[gsr_ssim_l1_scratch_floats(C,H,W)] (per-block
 *     partial sums, added up in a fixed order: the loss value is bitwise reproducible); optional ssim_map float[C,H,W];
 *     optional dm_dmu1, dm_dsigma1_sq, dm_dsigma12 float[C,H,W] (all three or none): planes saved for the backward.
 *   backward: weights float[2] (device) = { dL/d sums[0], dL/d sums[1] };
 *     dL_dimg1 float[C,H,W] <- weights[0] * sign(x - y) + weights[1] * d(sum ssim)/dx.  img2 gets no gradient. */
size_t gsr_ssim_l1_scratch_floats(int C, int H, int W);
int gsr_ssim_l1_forward(const float* img1, const float* img2, int C, int H, int W, float C1, float C2, float* sums,
                        float* scratch, float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                        void* stream);
int gsr_ssim_l1_backward(const float* img1, const float* img2, int C, int H, int W, const float* weights,
                         const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1,
                         void* stream);

/* gsr_surface_forward / _backward: the per-pixel chain render() runs after the rasterizer,
 * gaussian_renderer/__init__.py:151-176 + utils/point_utils.py:9-37 (depths_to_points, depth_to_normal):
 *   surf_depth  = nan_to_num(allmap[0] / clamp(allmap[1], 1e-3)) * (1 - depth_ratio) + depth_ratio * nan_to_num(allmap[5])
 *   surf_normal = normalize(cross(P[y+1] - P[y-1], P[x+1] - P[x-1])) * allmap[1].detach(), P = surf_depth * rays_d + rays_o,
 *                 zero on the one-pixel border.
 *   allmap float[8,H,W] (the rasterizer's out_others); raymat DEVICE float[12]: rays_d(x,y) = (x, y, 1) . M with M the
 *   first nine floats row-major (= intrins^-1.T @ c2w[:3,:3].T of depths_to_points), then rays_o[3];
 *   surf_depth float[H,W], surf_normal float[3,H,W].
 *   backward: surf_depth = the forward's output; g_surf_depth float[H,W] and g_surf_normal float[3,H,W] may each be NULL
 *   (= zero); g_allmap float[8,H,W] is fully written (planes 0, 1, 5 carry gradient, the rest zeros). */
int gsr_surface_forward(const float* allmap, const float* raymat, float depth_ratio, int H, int W, float* surf_depth,
                        float* surf_normal, void* stream);
int gsr_surface_backward(const float* allmap, const float* raymat, float depth_ratio, int H, int W,
                         const float* surf_depth, const float* g_surf_depth, const float* g_surf_normal, float* g_allmap,
                         void* stream);

/* Densification bookkeeping.
 * gsr_densification_stats: GaussianModel.add_densification_stats (scene/gaussian_model.py:578-584) fused with the
 *   max_radii2D update of train.py:243.  grad_means2D float[P,3] (viewspace_points.grad), radii int32[P]
 *   (update filter = radii > 0), gaussian_weights float[P]; the five float[P] statistics are updated in place.
 * gsr_gather_rows: applies one row map to several row-major blocks of a flat buffer: for every group g and new row r,
 *   dst[g.dst_offset + r*g.width + c] = row_map[r] >= 0 ? src[g.src_offset + row_map[r]*g.width + c] : 0.
 *   This is _prune_optimizer / cat_tensors_to_optimizer (scene/gaussian_model.py:403-484) for parameters and both Adam
 *   moments in ONE pass per buffer (new rows of the moments use map -1 = zeros, as torch.zeros_like there).
 *   row_map int32[n_rows] (device); groups: host array, at most 16; offsets in floats.
 * gsr_split_children: the new positions and scalings of densify_and_split (scene/gaussian_model.py:508-526):
 *   child j of parent[j]: xyz = R(q) (exp(scaling) * noise_j, 0) + xyz, scaling = log(exp(scaling) / (0.8 N)); noise
 *   float[n_children, scale_dims] standard normal supplied by the caller; scale_dims 2 (surfels) or 3. */
typedef struct {
	uint64_t src_offset, dst_offset;
	uint32_t width;
} gsr_gather_group;
int gsr_densification_stats(int P, const float* grad_means2D, const int* radii, const float* gaussian_weights,
                            float* xyz_gradient_accum, float* denom, float* accum_w, float* denom_w, float* max_radii2D,
                            void* stream);
int gsr_gather_rows(const float* src, float* dst, const int* row_map, uint64_t n_rows, const gsr_gather_group* groups,
                    int num_groups, void* stream);
int gsr_split_children(int n_children, int scale_dims, int N, const int* parent, const float* xyz, const float* scaling,
                       const float* rotation, const float* noise, float* child_xyz, float* child_scaling, void* stream);

/* Normal-consistency term of the training loss (reference train.py:182-189): normal_error = (1 - sum_c rend_normal[c] *
 * surf_normal[c]) [* mask], loss = lambda * mean(normal_error).  rend_normal, surf_normal: [3,H,W]; mask: [H,W] (1,H,W) or NULL.
 * Forward: sum2[0] = sum over pixels of normal_error (sum2[1] = 0), scratch = gsr_normal_loss_scratch_floats() floats.
 * Backward: g_sum = d loss / d sum2[0], ONE float in device memory (so that lambda / (H*W) and the upstream gradient need no
 * host synchronisation); writes g_rend_normal and g_surf_normal ([3,H,W]) fully; the mask gets no gradient. */
size_t gsr_normal_loss_scratch_floats(void);
int gsr_normal_loss_forward(const float* rend_normal, const float* surf_normal, const float* mask, int H, int W, float* sum2,
                            float* scratch, void* stream);
int gsr_normal_loss_backward(const float* rend_normal, const float* surf_normal, const float* mask, int H, int W,
                             const float* g_sum, float* g_rend_normal, float* g_surf_normal, void* stream);
/* gsr_adam_step: torch.optim.Adam(lr per group, betas, eps, amsgrad=False, weight_decay=0) as the reference sets it up
 * (scene/gaussian_model.py:196-209: eight groups, eps = 1e-15), fused over ONE flat buffer: param, grad, exp_avg and
 * exp_avg_sq are float[n], 16-byte aligned, laid out identically.  `segments` (host array, at most 16, tiling [0, n) in
 * order) give the learning rate of each parameter group; a group whose tensor interleaves two reference groups
 * (shs = cat(f_dc, f_rest): 3 of every 48 floats use feature_lr, the other 45 feature_lr / 20) sets period/split:
 * element i of the segment uses lr if ((i - begin) % period) < split else lr2; period 0 = plain lr.
 * `step` is the 1-based Adam step counter (bias corrections 1 - beta^step are formed in double on the host). */
typedef struct {
	uint64_t begin, end;
	float lr, lr2;
	uint32_t period, split;
} gsr_adam_segment;
int gsr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n,
                  const gsr_adam_segment* segments, int num_segments, float beta1, float beta2, float eps, int step,
                  void* stream);
/* Extension for the view-parallel step (no reference counterpart): the same update restricted to elements [range_begin, range_end)
 * of the SAME buffers and segment table (bounds multiples of 4; range_end may equal n).  A rank that received its 1/N of the
 * summed gradient by reduce-scatter steps only that shard (1/N of the optimizer's traffic and arithmetic) and the parameters are
 * all-gathered afterwards; gsr_adam_step is this call with the range [0, n). */
int gsr_adam_step_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n,
                        const gsr_adam_segment* segments, int num_segments, float beta1, float beta2, float eps, int step,
                        uint64_t range_begin, uint64_t range_end, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-stage device timing (bench.py's roofline leg).  When enabled, every stage launch is bracketed by
 * hipEvents recorded on the stream the stage runs on; gsr_profile_collect() synchronises those events,
 * adds up elapsed milliseconds and launch counts per stage since the last enable/collect, and resets.
 * Stage ids: */
#define GSR_STAGE_PREPROCESS 0
#define GSR_STAGE_SCAN 1       /* inclusive scan + 4-byte num_rendered readback */
#define GSR_STAGE_EMIT_KEYS 2
#define GSR_STAGE_SORT 3
#define GSR_STAGE_RANGES 4
#define GSR_STAGE_RENDER_FWD 5
#define GSR_STAGE_RENDER_BWD 6
#define GSR_STAGE_PREPROCESS_BWD 7
#define GSR_STAGE_REFL_FWD 8
#define GSR_STAGE_REFL_BWD 9
#define GSR_STAGE_CUBEMAP_FWD 10
#define GSR_STAGE_CUBEMAP_BWD 11
#define GSR_STAGE_LOSS_FWD 12
#define GSR_STAGE_LOSS_BWD 13
#define GSR_STAGE_ADAM 14
#define GSR_STAGE_SURFACE_FWD 15
#define GSR_STAGE_SURFACE_BWD 16
#define GSR_STAGE_REFL_BWD_TAIL 17   /* texel-gradient tail of the reflection backward (sort + run combine + unpack); on the side stream with async_tail */
#define GSR_STAGE_COUNT 18
/* Test/diagnostic switches.  "cull" (default 1): per-wave footprint culling inside the tile kernels (each wave votes
 * which list entries can reach its 8x8 pixel block at all); outputs are bit-identical with 0 and 1, it only skips
 * (wave, Gaussian) pairs that cannot blend.  "dev" (default 0): development ablation bits, not for production.  "emit_items" (default 0 =
 * chosen by the Gaussian count): 1 / 2 force the Gaussians per thread of key emission (tests).  "mailbox", "sort_driver" (default 1): 0 = the
 * round-2 read-back of num_rendered / the public rocPRIM sort entry points. */
int gsr_set_option(const char* name, int value);
int gsr_profile_enable(int on);
int gsr_profile_collect(float* ms_out /* [GSR_STAGE_COUNT] */, int* launches_out /* [GSR_STAGE_COUNT] */);

#ifdef __cplusplus
}
#endif
#endif /* GSR_HIP_H_ */
