// Blend kernels, default product arithmetic: fused multiply-adds (every one written out, see BlendMath<false>)
// + native exp2 / rcp.  Contraction is OFF for the translation unit: nothing else is fused behind our back.
#pragma clang fp contract(off)
#define FRG_EXACT false
#include "blend_impl.h"
#include "kernels.h"
namespace frg {
int g_bwd_tile_moments = 0;   // TIMING EXPERIMENT (frg_set_option("bwd_tile_moments"), FROSTING_EXPERIMENTS=1): round 3's moments about the tile centre
int g_fwd_prefetch = 1;   // frg_set_option("fwd_prefetch"): forward blend requests round r + 1's records before it processes round r
hipError_t launch_blend_fwd_fast(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
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

hipError_t launch_blend_bwd_fast(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                const float* bg, const float* dL_dpix, float* slots, int batch, int quad_tiles, hipStream_t s)
{
    const int T = vp.gx * vp.gy;
    // quad_tiles: at most this many active tiles -> the quadrant form (< 0: FRG_BWD_QUAD_TILES; 0: never)
    const uint32_t qt = quad_tiles < 0 ? (uint32_t)FRG_BWD_QUAD_TILES : (uint32_t)quad_tiles;
    hipLaunchKernelGGL(bwd_order_kernel, dim3(1), dim3(1024), 0, s, T, xcd_grid_blocks(T), img.tile_work, img.bwd_order, img.bwd_mode, qt, img.cutoff);
#define FRG_BWD(B, TM)                                                                                                         \
    hipLaunchKernelGGL((blend_bwd_kernel<FRG_EXACT, B, TM>), dim3(xcd_grid_blocks(T)), dim3(64), 0, s, T, vp.gx, vp.gy, vp.W, vp.H, \
                       img.ranges, b.point_list, g.xydr, g.conic_opacity, g.rgb_clamped, g.point_offsets, bg, img.final_T,     \
                       img.n_contrib, dL_dpix, slots, img.cutoff, img.bwd_order, img.bwd_mode)
    const bool tm = g_bwd_tile_moments != 0;
    if (tm) { if (batch == 2) FRG_BWD(2, true); else FRG_BWD(3, true); }
    else { if (batch == 2) FRG_BWD(2, false); else FRG_BWD(3, false); }
#undef FRG_BWD
    // the quadrant form for frames with few active tiles: one of the two launches finds the mode word against it and leaves
    const int nquad = (uint32_t)T < qt ? T : (int)qt;   // (inactive tiles sort behind the active ones)
#define FRG_BWDQ(B, TM)                                                                                                        \
    hipLaunchKernelGGL((blend_bwd_quad_kernel<FRG_EXACT, B, TM>), dim3(nquad), dim3(BLEND_THREADS), 0, s, T, vp.gx, vp.gy, vp.W, vp.H, \
                       img.ranges, b.point_list, g.xydr, g.conic_opacity, g.rgb_clamped, g.point_offsets, bg, img.final_T,     \
                       img.n_contrib, dL_dpix, slots, img.cutoff, img.bwd_order, img.bwd_mode)
    if (nquad > 0) {
        if (tm) { if (batch == 2) FRG_BWDQ(2, true); else FRG_BWDQ(3, true); }
        else { if (batch == 2) FRG_BWDQ(2, false); else FRG_BWDQ(3, false); }
    }
#undef FRG_BWDQ
    return hipGetLastError();
}
}  // namespace frg
