// Blend kernels, EXACT arithmetic: reference operation order, no FMA contraction,
// accurate expf.  Compiled with -ffp-contract=off as well.
#pragma clang fp contract(off)
#include "blend_impl.h"
#include "kernels.h"
namespace frg {
hipError_t launch_blend_fwd_exact(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                  const float* bg, float* out_color, hipStream_t s, bool forward_only, bool fused_sort, bool long_lists)
{
    return launch_blend_fwd_t<true>(vp, g, img, b, bg, out_color, g_fwd_prefetch != 0, s, forward_only, fused_sort, long_lists);
}

hipError_t launch_blend_bwd_exact(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                  const float* bg, const float* dL_dpix, float* slots, uint32_t R, int batch, hipStream_t s, bool as_stamped)
{
    return launch_blend_bwd_t<true>(vp, g, img, b, bg, dL_dpix, slots, R, batch, s, as_stamped);
}
}  // namespace frg
