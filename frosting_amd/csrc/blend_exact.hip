// Blend kernels, EXACT arithmetic: reference operation order, no FMA contraction,
// accurate expf.  Compiled with -ffp-contract=off as well.
#pragma clang fp contract(off)
#define FRG_EXACT true
#include "blend_impl.h"
#include "kernels.h"
namespace frg {
hipError_t launch_blend_fwd_exact(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                const float* bg, float* out_color, hipStream_t s)
{
    const int T = vp.gx * vp.gy;
    if (g_fwd_prefetch)
        hipLaunchKernelGGL((blend_fwd_kernel<FRG_EXACT, true>), dim3(xcd_grid_blocks(T)), dim3(BLEND_THREADS), 0, s, T, vp.gx, vp.W, vp.H,
                           img.ranges, b.point_list, g.xydr, g.conic_opacity, g.rgb_clamped, bg, img.final_T, img.n_contrib,
                           out_color, img.tile_work);
    else
    hipLaunchKernelGGL((blend_fwd_kernel<FRG_EXACT>), dim3(xcd_grid_blocks(T)), dim3(BLEND_THREADS), 0, s, T, vp.gx, vp.W, vp.H,
                       img.ranges, b.point_list, g.xydr, g.conic_opacity, g.rgb_clamped, bg, img.final_T, img.n_contrib,
                       out_color, img.tile_work);
    return hipGetLastError();
}

hipError_t launch_blend_bwd_exact(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                const float* bg, const float* dL_dpix, float* slots, int batch, int quad_tiles, hipStream_t s)
{
    const int T = vp.gx * vp.gy;
    // quad_tiles: at most this many active tiles -> the quadrant form (< 0: FRG_BWD_QUAD_TILES; 0: never)
    const uint32_t qt = quad_tiles < 0 ? (uint32_t)FRG_BWD_QUAD_TILES : (uint32_t)quad_tiles;
    hipLaunchKernelGGL(bwd_order_kernel, dim3(1), dim3(1024), 0, s, T, xcd_grid_blocks(T), img.tile_work, img.bwd_order, img.bwd_mode, qt, img.cutoff);
#define FRG_BWD(B)                                                                                                         \
    hipLaunchKernelGGL((blend_bwd_kernel<FRG_EXACT, B>), dim3(xcd_grid_blocks(T)), dim3(64), 0, s, T, vp.gx, vp.gy, vp.W, vp.H, \
                       img.ranges, b.point_list, g.xydr, g.conic_opacity, g.rgb_clamped, g.point_offsets, bg, img.final_T,     \
                       img.n_contrib, dL_dpix, slots, img.cutoff, img.bwd_order, img.bwd_mode)
    if (batch == 2) FRG_BWD(2); else FRG_BWD(3);
#undef FRG_BWD
    // the quadrant form for frames with few active tiles: one of the two launches finds the mode word against it and leaves
    const int nquad = (uint32_t)T < qt ? T : (int)qt;   // (inactive tiles sort behind the active ones)
#define FRG_BWDQ(B)                                                                                                        \
    hipLaunchKernelGGL((blend_bwd_quad_kernel<FRG_EXACT, B>), dim3(nquad), dim3(BLEND_THREADS), 0, s, T, vp.gx, vp.gy, vp.W, vp.H, \
                       img.ranges, b.point_list, g.xydr, g.conic_opacity, g.rgb_clamped, g.point_offsets, bg, img.final_T,     \
                       img.n_contrib, dL_dpix, slots, img.cutoff, img.bwd_order, img.bwd_mode)
    if (nquad > 0) { if (batch == 2) FRG_BWDQ(2); else FRG_BWDQ(3); }
#undef FRG_BWDQ
    return hipGetLastError();
}
}  // namespace frg
