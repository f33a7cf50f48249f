// TEST INFRASTRUCTURE (oracle side) -- not part of the shipped product.
//
// C-ABI wrapper around the *reference's own* rasterizer
// (DGR/cuda_rasterizer/rasterizer.h:20-85: CudaRasterizer::Rasterizer::
// {forward,backward,markVisible}), compiled by oracle/build_ref.sh from the
// sources where they lie under /root/reference into oracle/_ref/.  Nothing in
// this file restates the algorithm; it only forwards pointers and decodes the
// reference's opaque scratch chunks (rasterizer_impl.h:21-73) so that tests can
// compare intermediate artefacts (radii, tiles_touched, sort keys, point_list,
// ranges, n_contrib, final_T) bit for bit.
#include "rasterizer_impl.h"   // from the (sed-fixed, temporary) copy of DGR/cuda_rasterizer
#include <functional>
#include <cstring>

extern "C" {

typedef char* (*ref_alloc_fn)(void* ctx, size_t bytes);

int ref_forward(
    ref_alloc_fn geom_alloc, ref_alloc_fn binning_alloc, ref_alloc_fn img_alloc, void* ctx,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* opacities, const float* scales, float scale_modifier,
    const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered,
    float* out_color, int* radii, int debug)
{
    std::function<char*(size_t)> g = [=](size_t n) { return geom_alloc(ctx, n); };
    std::function<char*(size_t)> b = [=](size_t n) { return binning_alloc(ctx, n); };
    std::function<char*(size_t)> i = [=](size_t n) { return img_alloc(ctx, n); };
    int r = CudaRasterizer::Rasterizer::forward(g, b, i, P, D, M, background, width, height,
        means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
        viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered != 0, out_color, radii, debug != 0);
    return r;
}

void ref_backward(
    int P, int D, int M, int R, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug)
{
    CudaRasterizer::Rasterizer::backward(P, D, M, R, background, width, height, means3D, shs,
        colors_precomp, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
        campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix,
        dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
        dL_drot, debug != 0);
}

void ref_mark_visible(int P, float* means3D, float* viewmatrix, float* projmatrix, bool* present)
{
    CudaRasterizer::Rasterizer::markVisible(P, means3D, viewmatrix, projmatrix, present);
}

// Decoders for the opaque chunks: byte offsets of every named array inside the
// chunk the reference carved (GeometryState/ImageState/BinningState::fromChunk).
// out[] receives offsets relative to `chunk`.
void ref_geom_offsets(char* chunk, size_t P, long long* out)
{
    char* c = chunk;
    CudaRasterizer::GeometryState s = CudaRasterizer::GeometryState::fromChunk(c, P);
    out[0] = (char*)s.depths - chunk;
    out[1] = (char*)s.clamped - chunk;
    out[2] = (char*)s.internal_radii - chunk;
    out[3] = (char*)s.means2D - chunk;
    out[4] = (char*)s.cov3D - chunk;
    out[5] = (char*)s.conic_opacity - chunk;
    out[6] = (char*)s.rgb - chunk;
    out[7] = (char*)s.tiles_touched - chunk;
    out[8] = (char*)s.point_offsets - chunk;
}

void ref_image_offsets(char* chunk, size_t N, long long* out)
{
    char* c = chunk;
    CudaRasterizer::ImageState s = CudaRasterizer::ImageState::fromChunk(c, N);
    out[0] = (char*)s.accum_alpha - chunk;
    out[1] = (char*)s.n_contrib - chunk;
    out[2] = (char*)s.ranges - chunk;
}

void ref_binning_offsets(char* chunk, size_t R, long long* out)
{
    char* c = chunk;
    CudaRasterizer::BinningState s = CudaRasterizer::BinningState::fromChunk(c, R);
    out[0] = (char*)s.point_list - chunk;
    out[1] = (char*)s.point_list_unsorted - chunk;
    out[2] = (char*)s.point_list_keys - chunk;
    out[3] = (char*)s.point_list_keys_unsorted - chunk;
}

}  // extern "C"
