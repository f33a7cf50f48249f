/*
 * frosting_rasterizer.h -- C ABI of the MI355X-native differentiable Gaussian-splat
 * rasterizer (libfrosting_rasterizer.so, hand-written gfx950 HIP).
 *
 * This is the drop-in boundary.  Each entry point replaces one member of the
 * reference's native interface for this path
 *
 *   DGR = gaussian_splatting/submodules/diff-gaussian-rasterization
 *   DGR/cuda_rasterizer/rasterizer.h:20-85   CudaRasterizer::Rasterizer::{markVisible,forward,backward}
 *   DGR/rasterize_points.h:18-67             RasterizeGaussiansCUDA / ...BackwardCUDA / markVisible
 *   DGR/ext.cpp:15-18                        pybind exports rasterize_gaussians[_backward], mark_visible
 *
 * with the same argument order and meaning; the only additions are the HIP
 * stream (the reference launches on the legacy default stream), a user pointer
 * for the allocation callbacks (the reference passes std::function closures,
 * rasterize_points.cu:27-33) and an explicit workspace for backward.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float32 / int32 data unless noted;
 *   - optional inputs (shs | colors_precomp, scales+rotations | cov3D_precomp) are
 *     NULL when absent (the reference tests data_ptr()==nullptr, forward.cu:205,241);
 *   - matrices are the reference's row-vector 4x4 (world_view_transform,
 *     full_proj_transform), i.e. read column-major (auxiliary.h:58-77);
 *   - functions return >= 0 on success, a negative FRG_E* code on failure;
 *     frg_last_error() returns a thread-local message.  Without `debug` kernel
 *     faults are asynchronous, as in the reference (auxiliary.h:166-173);
 *   - no call allocates device memory behind the caller's back.
 */
#ifndef FROSTING_RASTERIZER_H_INCLUDED
#define FROSTING_RASTERIZER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRG_OK 0
#define FRG_EINVAL (-1)   /* bad argument (e.g. neither shs nor colors_precomp) */
#define FRG_EALLOC (-2)   /* an allocation callback returned NULL / too small workspace */
#define FRG_EHIP (-3)     /* a HIP runtime call or kernel failed (see frg_last_error) */
#define FRG_EFILTER (-4)  /* prefiltered=1 but a Gaussian was near-culled (auxiliary.h:154-162) */
#define FRG_ECAPACITY (-5) /* frg_forward_finish: the view had more instances than instance_capacity */

/* Resizable-buffer callback: must return device memory of at least `bytes`
 * bytes, 256-byte aligned, owned by the caller and kept alive until the matching
 * backward has run (replaces std::function<char*(size_t)>, rasterizer.h:32-34). */
typedef char* (*frg_alloc_fn)(void* user, size_t bytes);

/* API / ABI version of this header: 2.  (1 -> 2: the geometry chunk interleaves its three float4 arrays into one
 * 48-byte record per Gaussian -- frg_geometry_layout_n reports the stride; frg_forward_args / frg_backward_args grew
 * per-call modes, struct_size-gated; frg_backward_workspace_bytes grew by a hand-over list; everything version 1
 * callers could call keeps its signature.) */
#define FRG_VERSION 2
int frg_version(void);
const char* frg_last_error(void);

/* Replaces Rasterizer::markVisible (rasterizer.h:24-29, rasterizer_impl.cu:141-153):
 * present[i] = (view-space z > 0.2).  `present` is one byte per Gaussian. */
int frg_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     unsigned char* present, void* hip_stream);

/* Replaces Rasterizer::forward (rasterizer.h:31-56, rasterizer_impl.cu:198-336).
 * Returns num_rendered (= sum of tiles_touched) or a negative error code.
 * out_color is [3,H,W] planar and fully written; radii is [P] int32 and fully
 * written.  The three chunks obtained through the callbacks are opaque state for
 * frg_backward (layout: frg_*_layout below, for tests only).  One device->host
 * read of 48 bytes (num_rendered, sort work-list sizes) is the only host synchronisation. */
int frg_forward(frg_alloc_fn geometry_alloc, frg_alloc_fn binning_alloc, frg_alloc_fn image_alloc, void* user,
                int P, int D, int M,
                const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered,
                float* out_color, int* radii, int debug, void* hip_stream);

/* Forward without any host synchronisation, for training loops that keep the GPU queue full
 * (the reference blocks on a device->host read of num_rendered in the middle of the op,
 * rasterizer_impl.cu:280-281, because it sizes the binning buffer from it).  The caller states
 * an instance capacity instead: binning_alloc is asked once for frg_binning_bytes(capacity, INT_MAX)
 * and every launch that depends on the counters reads them on the device.  Returns
 * instance_capacity -- pass THAT as R to frg_backward_workspace_bytes / frg_backward -- or 0 when
 * P == 0, or a negative error.  Results are identical to frg_forward's.
 *
 * frg_forward_finish(image_buffer, ...) later waits for that forward's counters only (an event
 * behind the scan, not the whole stream) and reports the true num_rendered.  FRG_ECAPACITY: the view
 * had more instances than the capacity; nothing was rasterized (the image is the background, a
 * backward yields zeros) and the view must be repeated with a larger capacity.  Call it from the
 * thread that issued the forward, once per deferred forward that returned > 0. */
int frg_forward_deferred(frg_alloc_fn geometry_alloc, frg_alloc_fn binning_alloc, frg_alloc_fn image_alloc, void* user,
                         int P, int D, int M,
                         const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                         const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                         float tan_fovx, float tan_fovy, int prefiltered,
                         float* out_color, int* radii, int instance_capacity, void* hip_stream);
int frg_forward_finish(const char* image_buffer, int prefiltered, int* num_rendered);

/* Every forward option in one call.  Fields up to hip_stream mean what the frg_forward
 * parameters of the same name mean; set struct_size = sizeof(frg_forward_args).
 *   instance_capacity > 0   deferred counters (frg_forward_deferred); `debug` is then ignored.
 *   keep_mask != NULL       one byte per Gaussian; 0 = leave it out of this view exactly as if it had
 *                           been culled (radii 0, no instances, zero gradient rows).  This is Frosting's
 *                           mesh occlusion culling (frosting_scene/frosting_model.py:1564-1586) as a skip
 *                           flag: the reference compacts five per-Gaussian tensors with a boolean mask
 *                           before every render; the result for the kept Gaussians is bit-identical
 *                           (compaction preserves index order, hence the depth-tie order). */
typedef struct frg_forward_args {
    size_t struct_size;
    frg_alloc_fn geometry_alloc, binning_alloc, image_alloc;
    void* user;
    int P, D, M;
    const float* background;
    int width, height;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales;
    float scale_modifier;
    const float *rotations, *cov3D_precomp, *viewmatrix, *projmatrix, *cam_pos;
    float tan_fovx, tan_fovy;
    int prefiltered;
    float* out_color;
    int* radii;
    int debug;
    void* hip_stream;
    int instance_capacity;
    const unsigned char* keep_mask;
    /* ---- raw-parameter mode (optional; struct_size tells whether the caller knows these fields) ----------------
     * The model's RAW parameters instead of activated tensors: the activations the reference applies with eager
     * torch kernels every iteration run inside the per-Gaussian kernels and no activated tensor exists in memory.
     *   raw_opacities [P]   opacity  = sigmoid(raw)        (frosting_model.py:726-727, gaussian_model.py:104-106);
     *                       pass INSTEAD of opacities (which must then be NULL)
     *   raw_scales [P,3]    scale    = exp(raw)            (frosting_model.py:763, gaussian_model.py:96-98)
     *   raw_rotations [P,4] rotation = raw / max(|raw|, 1e-12)   (frosting_model.py:797-798); pass both INSTEAD of
     *                       scales / rotations
     *   shell_logits [P,6], shell_cell_verts [F,6,3], shell_cells [P] (int64): Frosting's shell-bound centres,
     *                       mean = softmax(logits) . the six vertices of the Gaussian's prismatic cell
     *                       (frosting_model.py:707-724); pass INSTEAD of means3D
     * Each group is optional on its own.  frg_backward_ex takes the same pointers. */
    const float *raw_opacities, *raw_scales, *raw_rotations;
    const float *shell_logits, *shell_cell_verts;
    const long long* shell_cells;
    /* ---- per-call modes (third generation of the struct; struct_size tells) ---------------------------------------
     * 0 = whatever frg_set_option says at the time of the call (the process-wide default), k + 1 = value k for THIS
     * call only: two rasterizers in two threads may hold different forward modes.  exact_blend 1 | 2 = fast | exact
     * blend arithmetic; tight_binning 1 | 2 = off | on; async_sh 1 .. 4 = modes 0 .. 3.
     *   shell_bary_mode  how shell_logits become barycentric weights (frosting_model.py:713-719): 0 softmax
     *                    (use_softmax_for_bary_coords = True, the default of the reference) | 1 relu + renormalise:
     *                    w = relu(x) / sum relu(x) -- torch.nn.functional.relu + the plain division the reference
     *                    writes (frosting_model.py:716-718), NO epsilon: a row whose six logits are all <= 0 gives
     *                    0 / 0 = NaN here exactly as it does there; gradient zero where x <= 0 */
    int exact_blend, tight_binning, async_sh;
    int shell_bary_mode;
    /* ---- fourth generation --------------------------------------------------------------------------------------
     * forward_only = 1: no backward will follow this forward (rendering, evaluation, torch.no_grad(): the reference's
     * autograd function keeps nothing either when no input needs a gradient, __init__.py:44-98).  The forward then leaves
     * out what it only writes for a backward -- the blend's checkpoints and final colours, the walked depths, cutoff keys
     * and work items of the tiles, and d(colour)/d(direction) of the SH pass: image, radii and the returned count are the
     * same bits.  frg_backward on its buffers is refused with FRG_EINVAL -- also on a copy of them at another address, and however
     * many forwards ago: the forward's blend kernel stamps "nothing kept" into the image chunk, and a backward whose host side
     * does not remember the forward reads that stamp.  Not offered with
     * instance_capacity > 0.  Forward alone, same process: C3 0.733 -> 0.696 ms, C2 0.0968 -> 0.0926 (the binning chunk is
     * sized as before: the checkpoints' 8 ... 16 bytes per instance are carved and left unwritten). */
    int forward_only;
} frg_forward_args;
int frg_forward_ex(const frg_forward_args* args);

/* Bytes of scratch frg_backward needs for a forward that returned R instances. */
size_t frg_backward_workspace_bytes(int P, int R);   /* slots (36 B per instance) + per-Gaussian sums (36 B per Gaussian) + one bit per Gaussian (phase 1: "has a gradient") */

/* Replaces Rasterizer::backward (rasterizer.h:58-84, rasterizer_impl.cu:340-434).
 * All nine gradient arrays are fully written (zero rows for culled Gaussians);
 * they need NOT be zero-initialised (the reference requires zeros,
 * rasterize_points.cu:151-159).  dL_dmean2D is [P,3] (z stays 0), dL_dconic
 * [P,4] = (a, b, -, c), dL_dopacity [P], dL_dcolor [P,3], dL_dmean3D [P,3],
 * dL_dcov3D [P,6], dL_dsh [P,M,3] (ignored when shs==NULL), dL_dscale [P,3],
 * dL_drot [P,4] (ignored when scales==NULL).  dL_dconic may be NULL: it is an intermediate that the
 * reference's binding never hands to Python (rasterize_points.cu:195).  Two more intermediates of the chain
 * may be NULL when the caller has no use for them (36 + 72 bytes per Gaussian less to write): dL_dcolor when shs and
 * dL_dsh are given, dL_dcov3D when the covariance comes from scales / rotations.  dL_dsh may be NULL with shs given: the
 * SH row is then not materialised (its view-direction term still reaches dL_dmean3D) and dL_dcolor
 * receives the clamp-masked colour gradient (backward.cu:31-34), i.e. the per-Gaussian factor dRGB of
 * dL_dsh[i][ch] = basis_i * dRGB[ch], from which frg_sh_grad_from_views rebuilds the row.  Summation order is fixed, so
 * results are bit-reproducible run to run (the reference's atomics are not).  The blend pass uses the arithmetic
 * (exact_blend) of the forward that filled the buffers -- see frg_backward_args::exact_blend.
 * geom_buffer is WRITTEN: one byte per Gaussian ("an instance of it was reached by a pixel's walk") that the next
 * forward on the buffer clears; a second backward on the same forward state finds the same marks.  Do not run two
 * backwards on one geometry buffer concurrently with a forward on it (the reference's buffers have the same rule). */
int frg_backward(int P, int D, int M, int R,
                 const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* campos,
                 float tan_fovx, float tan_fovy, const int* radii,
                 char* geom_buffer, char* binning_buffer, char* image_buffer,
                 const float* dL_dpix,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 char* workspace, size_t workspace_bytes, int debug, void* hip_stream);

/* frg_backward with every option, as a struct (set struct_size = sizeof(frg_backward_args)).  Fields up to
 * hip_stream mean what the frg_backward parameters of the same name mean.  In raw-parameter mode (the raw_* /
 * shell_* inputs of the forward, same pointers here) dL_dopacity, dL_dscale and dL_drot receive the gradients
 * with respect to the RAW parameters (sigmoid', exp', normalisation Jacobian applied); with shell-bound centres
 * dL_dshell_logits [P,6] is written (softmax Jacobian applied) and, when dL_dshell_cell_verts [F,6,3] is
 * non-NULL, the gradient of the cell vertices -- the learnable shell of learn_shell = True -- is ACCUMULATED
 * into it with float atomics (several Gaussians share a cell; the caller zeroes it).  dL_dmean3D is still written. */
typedef struct frg_backward_args {
    size_t struct_size;
    int P, D, M, R;
    const float* background;
    int width, height;
    const float *means3D, *shs, *colors_precomp, *scales;
    float scale_modifier;
    const float *rotations, *cov3D_precomp, *viewmatrix, *projmatrix, *campos;
    float tan_fovx, tan_fovy;
    const int* radii;
    char *geom_buffer, *binning_buffer, *image_buffer;
    const float* dL_dpix;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
    char* workspace;
    size_t workspace_bytes;
    int debug;
    void* hip_stream;
    const float *raw_opacities, *raw_scales, *raw_rotations;
    const float *shell_logits, *shell_cell_verts;
    const long long* shell_cells;
    float *dL_dshell_logits, *dL_dshell_cell_verts;
    /* second generation (struct_size tells): exact_blend 1 | 2 = fast | exact arithmetic of THIS backward's blend
     * pass; 0 (and frg_backward, which has no such argument) = the arithmetic of the forward that filled the buffers --
     * per-call mode or process default at the time of THAT call: the library remembers it for the 1024 most recent
     * geometry buffers of the process, and for a buffer it does not know (cloned, restored at another address, older) it
     * reads the word the forward's blend kernel stamped into image_buffer (one blocking 4-byte copy on hip_stream).
     * frg_set_option's value at the time of the backward plays no part.  A caller that carries the forward's mode beside
     * the buffers (the autograd ctx of frosting_amd/rasterizer.py does) states it here and saves that lookup.
     * shell_bary_mode as in frg_forward_args, and equal to the forward's */
    int exact_blend;
    int shell_bary_mode;
    /* third generation: the backward in TWO calls (0 = one call, everything).
     *   phase 1  the backward blend and the reduction of its per-instance partials to per-Gaussian sums (kept in the
     *            workspace); dL_dcolor is complete when this call's work is: with shs given and dL_dsh == NULL it holds
     *            the clamp-masked colour gradient -- the 12-byte-per-Gaussian payload of the factored view-parallel
     *            exchange, whose all-gather can then travel WHILE phase 2 runs;
     *   phase 2  everything else (the covariance / projection / SH chain and every other output), from those sums.
     * Same arguments and the same workspace in both calls; the two together write exactly what one call writes, bit for bit. */
    int phase;
    /* fourth generation: row_live (optional, [P] bytes).  NULL: every row of every gradient output is written, zero or
     * value (the reference's callers get that from their zero-fill, rasterize_points.cu:151-159).  Non-NULL: row_live[i] <- 1
     * for the Gaussians with a gradient in this view, 0 for the others -- visible or not -- and the rows of those others are
     * NOT written (at 3 M Gaussians six rows in seven are zeros: 0.74 of the 0.85 GB the backward writes).  A consumer that
     * treats an unmarked row as zero without reading it (frg_adam_step_rows) gets exactly what the dense form gives. */
    unsigned char* row_live;
    /* fifth generation: phase 1 IN PIECES (range_count > 0, with phase == 1 only; 0 = the whole phase in one call).  Every call
     * reduces the slots of Gaussians [range_first, range_first + range_count) -- range_first a multiple of 256 -- to their sums;
     * the call with range_first == 0 comes first and runs the backward blend in front of it.  The ranges together must cover
     * [0, P); each call's share of the sums (and of what frg_pack_sum_rows packs from them) is complete when that call's work
     * is, so a slot-sum exchange can pack and send range k while range k + 1 is still being reduced.  Same bits as one call. */
    int range_first, range_count;
} frg_backward_args;
int frg_backward_ex(const frg_backward_args* args);

/* Runtime options.  "exact_blend": 1 = blend kernels use the reference's IEEE
 * operation order without FMA contraction and the accurate expf (bit-identical
 * images to the reference built the same way); 0 (default) = FMA + native exp2.
 * The per-Gaussian stages are always evaluated in the exact order, so radii,
 * tile counts and sort keys never depend on this switch.  "profile": see
 * frg_stage_times; "profile_stage": k in 0..6 restricts the events to stage k (each event
 * record costs a few microseconds of stream time), -1 (default) = every stage.  "global_bins": 1 forces the binning path used for images with
 * more than 10112 tiles (global atomics instead of LDS histograms; test hook).
 * "tight_binning": 1 = a (Gaussian, tile) instance is only put on the tile's list if the Gaussian can
 * reach alpha >= 1/255 somewhere in the tile (the closed-form bound the blend kernels use per 8x8
 * quadrant, taken over the 16x16 tile); the reference lists every tile of the 3-sigma square
 * (forward.cu:236-255), about twice as many.  num_rendered, radii and the image are bit-identical either way; the
 * tile lists become order-preserving sub-lists of the reference's.  The gradients are bit-identical while no tile's
 * walk crosses a segment boundary of the backward blend ("bwd_seg_log"), and agree to float32 rounding beyond (a
 * boundary is a list position: the sub-list restarts the walk from another rounding of the same state).
 * Default 0: lists identical to the reference's, entry for entry.
 * "bwd_waves" (default 0 = one single-wave workgroup per work item, at most 16384): workgroups of the backward blend.
 * A work item is a SEGMENT of a tile's processed list prefix (the forward leaves every pixel's state at the segment
 * boundaries in the binning / image chunks, and lists the items as its tiles finish); the assignment of items to waves
 * is static, so the gradients do not depend on this number.  Scheduling only.
 * "bwd_seg_log" (default 0): log2 of the segment length.  0 = chosen per frame by the forward from the instance count its
 * binning chunk is carved for (256 entries below 2^23 instances, 512 from there: about as many items as the GPU holds
 * single-wave workgroups); 8, 9, 10 pin it.  Read by FORWARDS (it sizes the checkpoint space of frg_binning_bytes); a
 * backward follows what its forward stamped.  Gradients agree between lengths to float32 rounding of the checkpointed
 * state, not bit for bit.
 * "counter_mailbox" (default 1): the blocking forward learns num_rendered and the sort's class sizes from a pinned
 * host mailbox the scan workgroups post to with system-scope stores -- the scatter is enqueued while the scan stage
 * still runs -- instead of a copy + stream synchronisation behind the scan (0).  The binning callback is then asked
 * for frg_binning_bytes(num_rendered, 8193) (8193 = "longest tile list unknown": scratch of every sort path, as it is not
 * known yet), ~29 instead of 20 bytes per instance.  Not used with `debug`.  Same counters, same results.
 * "clear_image_state" (default 0): 1 = clear the image chunk's per-tile cursors and counters with a memset in front of
 * every forward even where the kernels initialise them on their way (images whose tiles fit the LDS bins).
 * "sparse_sh" (default 1): when a view is expected to see only a part of the model -- an occlusion mask (keep_mask) is
 * given, or the previous forward of the calling thread saw less than three quarters of a model of the same size -- the SH
 * pass streams the coefficient rows of the VISIBLE Gaussians only (0: always the rows of every 16-Gaussian block with a
 * visible one).  Same arithmetic per Gaussian: every output bit-identical.
 * "sh_dir_in_backward" (default 0): 1 = the forward's SH pass does not form d(colour)/d(direction) (36 bytes stored per
 * visible Gaussian, read back by the per-Gaussian backward); a backward that follows forms it from the 192-byte SH rows of
 * the Gaussians that have a gradient, in the forward's order of additions -- every gradient bit the same.  For forward-only
 * rendering of large models (C3: the per-Gaussian forward 0.221 -> 0.189 ms); with a backward behind it the step is SLOWER
 * (+0.064 ms in the per-Gaussian backward at C3), which is why it is not the default.  M = 16 coefficients only; the backward
 * follows what its forward stamped, not the option at the time it runs.
 * "fwd_order" (default 1): the forward blend takes the tiles longest list first (the size classes the scan builds for the
 * sort); 0 = in tile order, one band of tile rows per XCD.  Scheduling only, outputs bit-identical.
 * "fwd_prefetch" (default 1): the forward blend requests the next 64 list entries' records while it processes the
 * current ones (0: plain loop).  Scheduling only, outputs bit-identical.
 * "bwd_heavy_first" (default 1): when the forward posted that some 64-Gaussian waves own thousands of backward slots,
 * their 16-wave workgroups run on the caller's stream and the plain per-Gaussian kernel beside them on a side stream
 * (0: the other way round).  Scheduling only.
 * "sort_heavy_on_caller" (default 1): with tile lists beyond the LDS sort, their chain of kernels runs on the caller's
 * stream and the size classes on the side streams (0: the chain on a side stream).  Scheduling only.
 * Returns the previous value or FRG_EINVAL for an unknown name. */
int frg_set_option(const char* name, int value);
int frg_get_option(const char* name);

/* Per-stage GPU time of the calling thread's forwards/backwards since the previous call
 * (at most the last 64 launches of each stage), averaged per launch; valid after
 * frg_set_option("profile", 1): ms[0..6] = preprocess, scan, scatter, sort, blend_fwd,
 * blend_bwd, preprocess_bwd (milliseconds between hipEvents recorded on the caller's
 * stream; -1 when a stage did not run).  The events are only synchronised on here, so a
 * timed run of steps is not serialised by the measurement.  Returns the number of stages
 * (7) or a negative error. */
int frg_stage_times(float* ms, int n);

/* Sizes of the three state chunks (what the callbacks will be asked for).  max_tile_count: the longest tile list of the
 * view (lists beyond 8192 entries need a second pair buffer); 8193 stands for "unknown" and sizes for every sort path. */
size_t frg_geometry_bytes(int P);
size_t frg_image_bytes(int width, int height);
size_t frg_binning_bytes(int R, int max_tile_count);

/* Test-only introspection: byte offsets of the named arrays inside each chunk.
 * geometry: out[0]=xy_depth_radius (float4[P]: pixel x, pixel y, view depth, radius)
 *           out[1]=conic_opacity (float4[P]) out[2]=rgb_clamped (float4[P]: r,g,b, clamp bits)
 *           out[3]=tiles_touched (u32[P])    out[4]=point_offsets (u32[P], inclusive scan)
 *           out[5]=byte stride between consecutive Gaussians' float4 of out[0..2] (since library version 2 they are
 *           interleaved in one 48-byte record per Gaussian: element i of each lives at out[k] + i * out[5]);
 *           frg_geometry_layout_n only -- frg_geometry_layout keeps writing the five values version 1 callers
 *           made room for
 * image:    out[0]=final_T (f32[H*W]) out[1]=n_contrib (u32[H*W]) out[2]=ranges (uint2[tiles])
 *           out[3]=tile_count (u32[tiles])
 * binning:  out[0]=point_list (u32[R], sorted by (tile, depth, index))
 *           out[1]=pairs_unsorted (uint2[R]: depth bits, index; tile-major, unsorted) */
void frg_geometry_layout(int P, long long* out);               /* writes out[0..4] only (the form of library version 1) */
int frg_geometry_layout_n(int P, long long* out, int n);       /* writes min(n, 6) values, returns 6 = how many there are */
void frg_image_layout(int width, int height, long long* out);
void frg_binning_layout(int R, int max_tile_count, long long* out);

/* ---- triangle occlusion raster --------------------------------------------------
 * Replaces nvdiffrast's dr.rasterize as the reference uses it
 * (frosting_utils/nvdiffrast.py:53-54, frosting_utils/mesh_rasterization.py:146-156;
 * nvdiffrast itself is third-party and not in the reference tree).
 * pos: [V,4] clip-space vertices (= [v,1] @ full_proj_transform), tri: [F,3] int32,
 * rast: [H,W,4] float32 = (u, v, z/w, triangle_id + 1), all zeros where empty; u, v are
 * the perspective-correct barycentrics of vertices 0 and 1.  Pixel (col i, row j) is
 * sampled at NDC ((i+.5)/W*2-1, (j+.5)/H*2-1); -1 <= z/w <= 1; nearest z/w wins, ties go
 * to the smaller triangle id (deterministic). */
size_t frg_mesh_raster_workspace_bytes(int F, int width, int height);
int frg_mesh_rasterize(int V, int F, const float* pos, const int* tri, int width, int height,
                       float* rast, char* workspace, size_t workspace_bytes, void* hip_stream);
/* The same z-buffer pass reduced to what Frosting's occlusion culling consumes (frosting_model.py:1534-1539,
 * 1564-1566: pix_to_face -> the set of visible faces): face_visible[f] (one byte per triangle, fully written) = 1 iff
 * triangle f is the nearest surface at some pixel centre -- exactly the ids of frg_mesh_rasterize's fourth plane --
 * without the attribute resolve and the [H,W,4] plane (same workspace size). */
int frg_mesh_visible_faces(int V, int F, const float* pos, const int* tri, int width, int height,
                           unsigned char* face_visible, char* workspace, size_t workspace_bytes, void* hip_stream);

/* The per-frame occlusion culling of a Frosting refine step in one call (frosting_model.py:1524-1539 visible faces of the
 * shell's base mesh, :1564-1586 the per-Gaussian mask; frosting_utils/nvdiffrast.py:44-53 the clip-space vertices): from
 * the mesh's vertices verts [V,3], the camera's full_proj_transform (16 floats, row-major, applied as [v,1] @ M) and the
 * cell of every shell Gaussian (cell_of_point [n_shell], int64 as torch indexes) to
 *   face_visible [F]   as frg_mesh_visible_faces writes it,
 *   keep [n_shell + n_background] = face_visible[cell_of_point[i]] for the shell's Gaussians, 1 for the background ones --
 * the keep_mask of frg_forward_ex.  Five launches, no host work between them: vertex transform + the clears, the two
 * z-buffer passes, the marks, the gather (the composition of torch.ones / cat / matmul, frg_mesh_visible_faces and the
 * torch index it replaces is nine launches: 105 -> 45 us per C4 frame).  The clip-space positions are one fused
 * multiply-add per term in the order of the sum; the workspace also holds them (frg_mesh_occlusion_workspace_bytes). */
size_t frg_mesh_occlusion_workspace_bytes(int V, int F, int width, int height);
int frg_mesh_occlusion_mask(int V, int F, const float* verts, const float* full_proj_transform, const int* tri, int width, int height,
                            int n_shell, const long long* cell_of_point, int n_background, unsigned char* keep,
                            unsigned char* face_visible, char* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- view-parallel gradient exchange helpers ---------------------------------------
 * No counterpart in the (single-GPU) reference; SURVEY.md 8(e).  The per-view SH gradient
 * is rank one per Gaussian, dL_dsh_v[i][ch] = basis_i(normalize(mean - campos_v)) * dRGB_v[ch]
 * (backward.cu:20-139), with dRGB_v the colour gradient after the clamp mask (backward.cu:31-34),
 * so ranks exchange dRGB (3 floats) instead of dL_dsh (3*M floats) and rebuild the summed
 * SH gradient locally, term by term bit-identical to the per-view kernels.
 *
 * frg_sh_color_grad: out_drgb[P,3] = dL_dcolors * (clamped ? 0 : 1) using the clamp flags the
 * forward left in geom_buffer; rows with radii <= 0 are zero.
 * frg_sh_grad_from_views: dL_dsh[P,M,3] = sum_{v < n_views} basis(D, mean - campos_v) (x) drgb_v,
 * summed in view order; campos_v = campos + v*campos_stride, drgb_v = drgb + v*view_stride
 * (strides in floats).  Coefficients >= (D+1)^2 are written as zeros. */
int frg_sh_color_grad(int P, const char* geom_buffer, const int* radii, const float* dL_dcolors,
                      float* out_drgb, void* hip_stream);
int frg_sh_grad_from_views(int P, int D, int M, int n_views, const float* means3D,
                           const float* campos, long long campos_stride,
                           const float* drgb, long long view_stride, float* dL_dsh, void* hip_stream);

/* Sparse form of the exchange (round 5).  Of one view's per-Gaussian gradients only the rows of Gaussians some pixel reached
 * are non-zero (one visible Gaussian in seven at 3 M Gaussians), so what travels between ranks are ROWS of 16 floats
 *   { index (uint32 bit pattern), dL_dmeans3D[3], dL_dscales[3], dL_dopacity, dL_drotations[4], dRGB[3], 0 }
 * of the Gaussians with a non-zero row: 64 bytes per Gaussian with a gradient instead of 56 per Gaussian.
 * frg_pack_grad_rows: rows <- the non-zero rows of the five arrays, compacted (in no particular order; indices are unique);
 *   *count (device, 4 bytes) <- their number, which may exceed capacity_rows -- then only capacity_rows of them were
 *   written and the caller packs again into a larger buffer.  `rows` 16-byte aligned.
 * frg_scatter_grad_rows: adds n_rows rows INTO the four dense arrays (+=) and, when drgb_dense != NULL, stores their dRGB
 *   at drgb_dense[3 * index ..] (the layout frg_sh_grad_from_views reads).  One call per view, in view order, on one
 *   stream: every element then receives its terms in view order -- the single-process accumulation, bit for bit. */
int frg_pack_grad_rows(int P, const float* dL_dmeans3D, const float* dL_dscales, const float* dL_drotations, const float* dL_dopacity,
                       const float* drgb, float* rows, long long capacity_rows, unsigned int* count, void* hip_stream);
int frg_scatter_grad_rows(long long n_rows, int P, const float* rows, float* dL_dmeans3D, float* dL_dscales, float* dL_drotations,
                          float* dL_dopacity, float* drgb_dense, void* hip_stream);

/* Slot-sum form of the exchange (round 6).  After phase 1 of a two-call backward (frg_backward_args::phase = 1) everything a
 * view contributes to a Gaussian's gradient is determined by the nine per-Gaussian sums that call leaves in its workspace --
 * the clamp-masked colour gradient and six pixel moments -- together with the parameters and the view's camera, which every
 * rank holds.  Only the Gaussians a pixel reached have such sums (one in eight at 3 M Gaussians).  Ranks therefore exchange
 * PACKETS of those sums instead of finished gradients, and every rank runs the per-Gaussian chain of phase 2 itself, for
 * every view, in one pass:
 *   frg_pack_sum_rows     (after phase 1, same P / R / workspace; drgb_masked = the dL_dcolor that call wrote with shs given
 *                         and dL_dsh == NULL) writes the packet of Gaussians [first, first + count), first a multiple of 64:
 *                         a 256-byte header (rows wanted / packed, the camera), one bit per Gaussian, one row offset per 64
 *                         Gaussians, and the 48-byte rows {dRGB[3], moments[6], view-direction terms[3]} of the marked Gaussians IN INDEX ORDER, at
 *                         most capacity_rows of them.  Fixed size frg_sum_packet_bytes(count, capacity_rows): it can be
 *                         all-gathered without any host knowing a count.  Header word 1 > capacity_rows: overflow, the
 *                         packet is incomplete (pack again with a larger capacity; the workspace is untouched).
 *   frg_backward_combine  n_views packets of the same range (packets + v * packet_stride_bytes, as an all-gather leaves them)
 *                         -> dL_dmean3D, dL_dscale, dL_drot, dL_dopacity, dL_dsh of Gaussians [first, first + count): the sum
 *                         over the views, IN VIEW ORDER, of what frg_backward phase 2 writes for each view -- every row
 *                         written once (zeros where no view has a row; with row_live != NULL those are not written and the
 *                         byte says so, as frg_backward_args::row_live).  Same expressions without contraction and the same
 *                         order of additions as accumulating the per-view gradients in one process: the same bits.  M = 16
 *                         coefficients; raw opacity / scale / rotation inputs as in frg_backward_ex (their Jacobians are applied
 *                         per view, as there); shell-bound centres are not offered.
 *                         status (optional; device or pinned host memory, 1 + n_views 64-bit words, each written at once as
 *                         status_seq << 32 | value): word 0 <- 1 if any packet overflowed its capacity or does not describe
 *                         this range, word 1 + v <- the rows view v wanted.  A host polling pinned memory until every word
 *                         carries status_seq learns the verdict while the pass runs, without synchronising the stream (the
 *                         pass's first workgroup posts it as it starts).  On overflow the outputs are incomplete. */
size_t frg_sum_packet_bytes(int n_gaussians, long long capacity_rows);
int frg_pack_sum_rows(int P, int R, int first, int count, char* workspace, size_t workspace_bytes, const float* drgb_masked,
                      const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                      int width, int height, float scale_modifier, int D, void* packet, size_t packet_bytes, long long capacity_rows,
                      void* hip_stream);
typedef struct frg_combine_args {
    size_t struct_size;
    int P, first, count, n_views;
    const void* packets;
    size_t packet_stride_bytes;
    long long capacity_rows;
    int M;
    const float *means3D, *shs, *scales, *rotations, *opacities;
    const float *raw_opacities, *raw_scales, *raw_rotations;
    float *dL_dmean3D, *dL_dscale, *dL_drot, *dL_dopacity, *dL_dsh;
    unsigned long long* status;
    unsigned int status_seq;
    unsigned char* row_live;
    char* workspace;              /* frg_combine_workspace_bytes(n_views, capacity_rows): 256 bytes (the pass is one kernel and stages nothing in HBM) */
    size_t workspace_bytes;
    void* hip_stream;
} frg_combine_args;
size_t frg_combine_workspace_bytes(int n_views, long long capacity_rows);
int frg_backward_combine(const frg_combine_args* args);

/* ---- fused Adam over the flat per-Gaussian parameter layout ---------------------------
 * SURVEY.md 8(f) rank 1, the step right after the backward / the gradient exchange.  Replaces
 * torch.optim.Adam(groups, lr=0.0, eps=1e-15).step() as the reference sets it up
 * (frosting_scene/frosting_optimizer.py:74-121, gaussian_splatting/scene/gaussian_model.py:149-167):
 * one parameter group per tensor with its own learning rate, betas (0.9, 0.999), no weight decay,
 * no amsgrad.  params / grads / exp_avg / exp_avg_sq are flat fp32 arrays of n elements with the
 * same layout (16-byte aligned); segment k covers [segment_ends[k-1], segment_ends[k]) and uses
 * segment_lrs[k]; the last segment must end at n.  Optionally (all three arrays non-NULL, period > 0)
 * a segment has a periodic head: elements whose offset in the segment modulo segment_period[k] is below
 * segment_head[k] use segment_head_lrs[k] -- one [P,16,3] SH tensor is then stepped as the reference's two
 * groups features_dc (lr) and features_rest (lr/20) without the torch.cat of every iteration
 * (gaussian_model.py:get_features, frosting_model.py:sh_coordinates).  `step` is the 1-based step count of the bias
 * corrections, grad_scale multiplies the gradient first (1/world for a mean over views).  One
 * launch, 28 bytes of HBM traffic per element; arithmetic as torch's _single_tensor_adam. */
#define FRG_ADAM_MAX_SEGMENTS 8
int frg_adam_step(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                  const long long* segment_ends, const float* segment_lrs, const int* segment_period,
                  const int* segment_head, const float* segment_head_lrs, int n_segments,
                  double beta1, double beta2, double eps, int step, float grad_scale, void* hip_stream);

/* ---- nearest-neighbour distances (scale initialisation) --------------------------------------
 * SURVEY.md 8(f) rank 4.  Replaces simple-knn's distCUDA2 (gaussian_splatting/submodules/simple-knn/
 * simple_knn.cu:64-222, spatial.cu:15-26; callers gaussian_model.py:134, frosting_model.py:530):
 * mean_dist2[i] = mean of the three smallest squared distances from points[i] ([P,3] float32) to the
 * other points, (b0 + b1 + b2) / 3 with b ascending -- exact search, the reference's values. */
size_t frg_knn_workspace_bytes(int P);
int frg_knn_mean_dist2(int P, const float* points, float* mean_dist2, char* workspace, size_t workspace_bytes,
                       void* hip_stream);

/* ---- Frosting shell parameterisation of the centres -------------------------------------------
 * The rest of SURVEY.md 8(f) rank 3.  points = sum_k softmax(bary_logits)[k] * cell_verts[cell][k]
 * (frosting_scene/frosting_model.py:713-724; cell_verts [F,6,3] = shell_cells_verts.reshape(-1, 6, 3),
 * the inner then the outer triangle of each prismatic cell, constants when learn_shell = False;
 * point_cell_indices [P] int64 as the reference stores them) and its backward w.r.t. the logits. */
int frg_shell_points(int P, const float* bary_logits, const float* cell_verts, const long long* point_cell_indices,
                     float* points, void* hip_stream);
int frg_shell_points_backward(int P, const float* bary_logits, const float* cell_verts, const long long* point_cell_indices,
                              const float* dL_dpoints, float* dL_dlogits, void* hip_stream);

/* ---- parameter activations -----------------------------------------------------------------
 * Part of SURVEY.md 8(f) rank 3.  opacity = sigmoid(raw), scale = exp(raw), rotation = F.normalize(raw)
 * (gaussian_splatting/scene/gaussian_model.py:32-40,96-115; frosting_scene/frosting_model.py:32,726,797-798)
 * in one launch; frg_activate_backward turns, IN PLACE, the rasterizer's gradients w.r.t. the activated
 * values (g_opacity [P], g_scale [P,3], g_rot [P,4]) into gradients w.r.t. the raw parameters, so the flat
 * gradient buffer can go straight into frg_adam_step. */
int frg_activate(int P, const float* raw_opacity, const float* raw_scale, const float* raw_rot,
                 float* opacity, float* scale, float* rot, void* hip_stream);
int frg_activate_backward(int P, const float* opacity, const float* scale, const float* raw_rot,
                          float* g_opacity, float* g_scale, float* g_rot, void* hip_stream);

/* frg_adam_step on a SHARD of the flat layout (reduce-scatter of the gradients -> every rank updates its 1/N of
 * the parameters -> all-gather of the parameters): elements [first, first + n), first a multiple of 4; the four arrays point at
 * element `first` (the moments may be shard-sized allocations); segment_* describe the WHOLE layout as for frg_adam_step, at
 * most 7 segments.  Every element gets the step size of its true segment and phase: the shards' updates together are
 * frg_adam_step's on the whole buffer, bit for bit. */
int frg_adam_step_shard(long long n, long long first, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                        const long long* segment_ends, const float* segment_lrs, const int* segment_period,
                        const int* segment_head, const float* segment_head_lrs, int n_segments,
                        double beta1, double beta2, double eps, int step, float grad_scale, void* hip_stream);

/* frg_adam_step with a row mask: row_live[P] as frg_backward_args::row_live leaves it, segment_width[k] = elements per
 * Gaussian of segment k (0: the segment is not per-Gaussian -- its gradients are always read).  The gradient of an
 * element whose Gaussian is unmarked is taken as zero WITHOUT being read (the moments still decay and the parameter still
 * moves by its momentum: the reference's dense semantics); everything else as frg_adam_step.  Bit-identical to
 * frg_adam_step on the dense gradient.  A per-Gaussian segment whose rows are a multiple of 4 elements long must BEGIN on
 * a multiple of 4 elements (FRG_EINVAL otherwise: the kernel reads one mask byte per aligned group of four there); the
 * 16-byte aligned segments of frosting_amd.parallel.flat_layout satisfy this. */
int frg_adam_step_rows(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                       const long long* segment_ends, const float* segment_lrs, const int* segment_period,
                       const int* segment_head, const float* segment_head_lrs, int n_segments,
                       double beta1, double beta2, double eps, int step, float grad_scale,
                       const unsigned char* row_live, int P, const int* segment_width, void* hip_stream);

/* ---- fused photometric loss (forward + backward) ----------------------------------------
 * SURVEY.md 8(f) rank 2, the step right before the rasterizer's backward.  Replaces
 *     (1 - lambda) * l1_loss(image, target) + lambda * (1 - ssim(image, target))
 * of frosting_utils/loss_utils.py:17-62 as frosting_trainers/refine.py:407-409 uses it (lambda = 0.2),
 * and its autograd backward: image, target [C,H,W] planar float32; window11 = the 11 taps of the
 * reference's normalised 1-D Gaussian window (loss_utils.py:23-25; HOST memory, the 2-D window there is
 * its outer product); loss (device, 1 float) and dL_dimage [C,H,W] = d loss / d image are written
 * (dL_dimage may be NULL for the value only).  Zero padding like conv2d(padding=5); the scalar is
 * reduced in a fixed order (bit-reproducible).  workspace: frg_photometric_workspace_bytes. */
size_t frg_photometric_workspace_bytes(int channels, int width, int height);
int frg_photometric_loss(int channels, int width, int height, const float* image, const float* target,
                         const float* window11, float lambda_dssim, float* loss, float* dL_dimage,
                         char* workspace, size_t workspace_bytes, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
