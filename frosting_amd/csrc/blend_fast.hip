// Blend kernels, default product arithmetic: fused multiply-adds (every one written out, see BlendMath<false>)
// + native exp2 / rcp.  Contraction is OFF for the translation unit: nothing else is fused behind our back.
#pragma clang fp contract(off)
#include "blend_impl.h"
#include "kernels.h"
namespace frg {
int g_bwd_waves = 0;
int g_fwd_order = 1;   // frg_set_option("fwd_order"): 1 = the forward blend takes the tiles longest list first, 0 = XCD band by band (rounds 1-3)
int g_fwd_prefetch = 1;   // frg_set_option("fwd_prefetch"): forward blend requests round r + 1's records before it processes round r
hipError_t launch_blend_fwd_fast(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                 const float* bg, float* out_color, hipStream_t s, bool forward_only, bool fused_sort, bool long_lists)
{
    return launch_blend_fwd_t<false>(vp, g, img, b, bg, out_color, g_fwd_prefetch != 0, s, forward_only, fused_sort, long_lists);
}

hipError_t launch_blend_bwd_fast(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                 const float* bg, const float* dL_dpix, float* slots, uint32_t R, int batch, hipStream_t s, bool as_stamped)
{
    return launch_blend_bwd_t<false>(vp, g, img, b, bg, dL_dpix, slots, R, batch, s, as_stamped);
}
}  // namespace frg
