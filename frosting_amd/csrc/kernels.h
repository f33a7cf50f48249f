// Host-callable launchers, one per kernel translation unit (each TU is compiled
// with the floating-point contraction mode its arithmetic contract needs).
#pragma once
#include "frg_common.h"
#include "raw_params.h"

namespace frg {

struct FwdInputs {
    const float *means3D, *scales, *rotations, *opacities, *shs, *cov3D_precomp, *colors_precomp;
    const float *viewmatrix, *projmatrix, *cam_pos;
    const unsigned char* keep_mask = nullptr;   // optional per-Gaussian skip flag (0 = not in this view)
    RawInputs raw;                              // optional: the model's raw parameters instead of activated tensors
};

// defer_sh: the SH colours are left to launch_sh_color (any stream ordered after this launch, before the blend)
hipError_t launch_preprocess_fwd(int P, const ViewParams& vp, const FwdInputs& in, int* radii, const GeomState& g,
                                 const ImageState& img, int prefiltered, bool defer_sh, hipStream_t s);
hipError_t launch_sh_color(int P, const ViewParams& vp, const FwdInputs& in, const int* radii, const GeomState& g, hipStream_t s);
// mail (optional, pinned host memory): where the scan workgroups post the counters for the polling host thread
hipError_t launch_scan(int P, const ViewParams& vp, const GeomState& g, const ImageState& img, uint32_t capacity, hipStream_t s,
                       Mailbox* mail = nullptr, uint32_t seq = 0);
hipError_t launch_scatter(int P, const ViewParams& vp, const int* radii, const GeomState& g, const ImageState& img,
                          const BinningState& b, hipStream_t s, int ablate = 0, Mailbox* mail = nullptr, uint32_t seq = 0);
extern int g_rows_grid;   // workgroups of the row-ordered scatter (tuning)
hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, unsigned char* present, hipStream_t s);

// class_count: host copy of the per-class tile counts, or nullptr when they are only known on the
// device (deferred-counters forward) -- grid_hint[] then sizes the launches and the workgroups
// stride over the device-side lists (class_count_dev), whatever their true length.
// big_plan / big_hist: BinningState's scratch of the two sorts of lists longer than the LDS capacity (R: the instance
// count the binning chunk was carved with); max_tile_count: longest list (0: unknown, kernels stride); index_bits:
// bits needed for a Gaussian index (tie order = ascending index)
// to be called before launch_scatter (same arguments as launch_tile_sort's): plans the sort of the long lists
extern int g_sort_heavy_on_caller;
extern int g_fwd_prefetch;
extern int g_fwd_order;
extern int g_bwd_waves;         // tuning: single-wave workgroups of the backward blend (0: the default, 16 per CU)
hipError_t launch_sort_plan(int T, const uint32_t* class_count, const uint32_t* class_count_dev, const uint32_t* class_tiles,
                            const uint2* ranges, uint32_t* big_plan, uint32_t R, hipStream_t stream, int fork_mode = 0);
hipError_t launch_tile_sort(int T, const uint32_t* class_count, const uint32_t* grid_hint, const uint32_t* class_count_dev,
                            const uint32_t* class_tiles, const uint2* ranges, uint2* pairs, uint2* pairs_tmp,
                            uint32_t* big_hist, uint32_t* big_plan, uint32_t R, int max_tile_count, int index_bits,
                            uint32_t* point_list, hipStream_t stream, bool skip_small = false /* the (0, 512] class is left to the forward blend's fused form */);

// forward_only: no backward will follow (frg_forward_args::forward_only) -- no checkpoints, no final colours, no work items
// fused_sort: the lists of at most 512 entries are sorted by the blend's own workgroups (launch_tile_sort was told skip_small)
// long_lists: the frame's work sits in a few long lists (the host's reading of the counters): eight list entries per trip instead of four
hipError_t launch_blend_fwd_exact(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                  const float* bg, float* out_color, hipStream_t s, bool forward_only = false, bool fused_sort = false, bool long_lists = false);
hipError_t launch_blend_fwd_fast(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                 const float* bg, float* out_color, hipStream_t s, bool forward_only = false, bool fused_sort = false, bool long_lists = false);
// batch: instances reduced together per step of the backward blend (2 or 3; tuning knob, same results up to rounding order)
// R: the instance count the caller sized the slots for (bounds the number of work items: the grid)
// as_stamped: the host launches BOTH arithmetics and each kernel leaves at once unless the forward's stamp (Counters::fwd_flags)
// names it; either way a forward_only stamp leaves the launch without work
hipError_t launch_blend_bwd_exact(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                  const float* bg, const float* dL_dpix, float* slots, uint32_t R, int batch, hipStream_t s, bool as_stamped = false);
hipError_t launch_blend_bwd_fast(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                 const float* bg, const float* dL_dpix, float* slots, uint32_t R, int batch, hipStream_t s, bool as_stamped = false);

struct BwdOutputs {
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
    // raw-parameter mode (FwdInputs::raw): dL_dopacity / dL_dscale / dL_drot then receive the gradients w.r.t. the RAW
    // parameters; with shell-bound centres dL_dshell_logits [P,6] is written and, when non-NULL, dL_dshell_verts
    // [F,6,3] is ACCUMULATED into (caller zeroes it): the learnable shell of learn_shell = True
    float *dL_dshell_logits = nullptr, *dL_dshell_verts = nullptr;
    // optional ([P] bytes): 1 = the Gaussian has a gradient in this view; the rows of the others are then NOT written
    unsigned char* row_live = nullptr;
};
// heavy_only: false = every wave of 64 Gaussians that is not on GeomState::heavy_waves (the plain kernel), true = the
// listed waves (the 16-wave form; any stream ordered after the blend backward).  flags:
#define FRG_PBW_NO_HEAVY_LAUNCH 2  // the 16-wave launch is skipped: the plain kernel reduces waves of any slot count itself
#define FRG_PBW_SUMS_ONLY 4        // phase 1 of a two-call backward: reduce the slots, store the nine sums per Gaussian (`sums`) and dL_dcolor, stop
#define FRG_PBW_FROM_SUMS 8        // phase 2: take the nine sums from `sums` instead of reducing the slots; dL_dcolor is already written
// live_masks (phase 1, optional): one bit per Gaussian -- "its nine sums are not all zero" -- as 64-bit words per wave of 64
hipError_t launch_preprocess_bwd(int P, const ViewParams& vp, const FwdInputs& in, const int* radii, const GeomState& g,
                                 const ImageState& img, const float* slots, const BwdOutputs& out, int ablate, int flags,
                                 bool heavy_only, hipStream_t s, float* sums = nullptr, unsigned long long* live_masks = nullptr,
                                 float* view_dir_terms = nullptr /* [P][3], with live_masks: d(colour)/d(direction) . masked dRGB of the marked Gaussians */,
                                 int range_first = 0, int range_count = 0 /* > 0 (plain kernel only): Gaussians [range_first, + range_count), range_first a multiple of 256 */);

// view-parallel exchange helpers (view_exchange.hip)
hipError_t launch_sh_color_grad(int P, const GeomState& g, const int* radii, const float* dL_dcolor, float* out, hipStream_t s);
hipError_t launch_sh_grad_from_views(int P, int D, int M, int n_views, const float* means3D, const float* campos,
                                     long long campos_stride, const float* drgb, long long view_stride, float* dL_dsh,
                                     hipStream_t s);

// sparse form of the exchange: rows (index, 11 dense floats, dRGB; 16 floats) of the Gaussians with a gradient
hipError_t launch_pack_grad_rows(int P, const float* g_means3D, const float* g_scales, const float* g_rot, const float* g_opac,
                                 const float* drgb, float* rows, unsigned int capacity, unsigned int* count, hipStream_t s);
hipError_t launch_scatter_grad_rows(unsigned int n, int P, const float* rows, float* g_means3D, float* g_scales, float* g_rot,
                                    float* g_opac, float* drgb_dense, hipStream_t s);

// slot-sum exchange (slot_exchange.hip): rows of the nine per-Gaussian sums of phase 1, packed in index order behind a bit mask;
// one combine pass runs the per-Gaussian chain for every view's row in view order
#define FRG_SUM_HDR_WORDS 64
#define FRG_SUM_ROW_FLOATS 12     // masked dRGB[3], six pixel moments, three view-direction terms: 48 bytes, three aligned float4
#define FRG_SUM_MAGIC 0x46534d36u
struct SumCamera { float tan_fovx, tan_fovy, scale_modifier; int width, height, D; };
size_t sum_packet_bytes(size_t n, size_t capacity);
// live_masks: one bit per Gaussian, as phase 1 of the backward leaves them in its workspace; sums: [P][9] of the same workspace
hipError_t launch_pack_sum_rows(int first, int n, uint32_t capacity, const unsigned long long* live_masks, const float* sums,
                                const float* view_dir_terms, const float* drgb_masked, const SumCamera& cam, const float* viewmatrix, const float* projmatrix,
                                const float* campos, void* packet, uint32_t* group_tot /* scratch: one word per 16384 Gaussians */, hipStream_t s);
// in: means3D, shs, scales, rotations, opacities (or their raw forms); out: dL_dmean3D, dL_dscale, dL_drot, dL_dopacity, dL_dsh
// workspace: combine_workspace_bytes(n_views, capacity) (256 bytes since the pass became one kernel that stages nothing in HBM);
size_t combine_workspace_bytes(int n_views, size_t capacity);
extern int g_combine_blocks;   // tuning: blocks of 64 Gaussians per combine tile (0: by the number of views)
hipError_t launch_backward_combine(int first, int n, int n_views, const void* packets, size_t packet_stride_bytes, uint32_t capacity,
                                   const FwdInputs& in, const BwdOutputs& out, unsigned long long* status, uint32_t seq, unsigned char* row_live,
                                   char* workspace, hipStream_t s);

// fused Adam over the flat parameter layout (adam.hip)
#ifndef FRG_ADAM_MAX_SEGMENTS
#define FRG_ADAM_MAX_SEGMENTS 8
#endif
struct AdamSegments {
    long long end[FRG_ADAM_MAX_SEGMENTS];    // exclusive end of segment k (begin = end of k-1), in elements
    float step_size[FRG_ADAM_MAX_SEGMENTS];  // lr_k / (1 - beta1^t)
    // optional sub-structure of a segment: elements whose (offset in segment) % period < head use
    // head_step_size instead (one [P,16,3] SH tensor optimised as the reference's two groups,
    // features_dc and features_rest, without ever concatenating them); period 0 = none
    int period[FRG_ADAM_MAX_SEGMENTS], head[FRG_ADAM_MAX_SEGMENTS];
    float head_step_size[FRG_ADAM_MAX_SEGMENTS];
    int count;
};
// row_live (optional, with width[k] = elements per Gaussian of segment k, 0 = not per-Gaussian, P Gaussians): the gradient
// of an element whose Gaussian is unmarked is zero and is not read
struct AdamRows {
    const unsigned char* live = nullptr;
    int P = 0;
    int width[FRG_ADAM_MAX_SEGMENTS] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int magic[FRG_ADAM_MAX_SEGMENTS] = {0, 0, 0, 0, 0, 0, 0, 0};     // floor(2^32 / width): offset / width without a division
};
hipError_t launch_adam_step(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                            const AdamSegments& seg, float w1, float beta2, float omb2, float inv_bc2_sqrt, float eps,
                            float grad_scale, hipStream_t s, const AdamRows* rows = nullptr);

// Frosting shell parameterisation of the centres (shell.hip)
hipError_t launch_shell_points(int P, const float* logits, const float* cell_verts, const long long* cell, float* points,
                               hipStream_t s);
hipError_t launch_shell_points_bwd(int P, const float* logits, const float* cell_verts, const long long* cell,
                                   const float* dL_dpoints, float* dL_dlogits, hipStream_t s);

// parameter activations (activations.hip)
hipError_t launch_activate(int P, const float* raw_opacity, const float* raw_scale, const float* raw_rot, float* opacity,
                           float* scale, float* rot, hipStream_t s);
hipError_t launch_activate_bwd(int P, const float* opacity, const float* scale, const float* raw_rot, float* g_opacity,
                               float* g_scale, float* g_rot, hipStream_t s);

// 3-nearest-neighbour mean squared distance (knn.hip)
size_t knn_workspace_bytes(int P);
hipError_t launch_knn(int P, const float* pts, float* out, char* workspace, hipStream_t s);

// fused photometric loss (photometric.hip)
size_t photometric_workspace_bytes(int C, int W, int H);
hipError_t launch_photometric(int C, int W, int H, const float* pred, const float* gt, const float* window11, float lambda,
                              float* loss, float* dL_dpred, char* workspace, hipStream_t s);

}  // namespace frg
