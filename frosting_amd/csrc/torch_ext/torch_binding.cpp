// PyTorch-ROCm extension module `diff_gaussian_rasterization._C`.
//
// Drop-in for the reference's pybind module (DGR = gaussian_splatting/submodules/
// diff-gaussian-rasterization): DGR/ext.cpp:15-18 exports rasterize_gaussians,
// rasterize_gaussians_backward and mark_visible with the signatures of
// DGR/rasterize_points.h:18-67; so does this file, positional argument for positional
// argument and tuple slot for tuple slot.  Underneath there is no CUDA-shaped code at all:
// the three functions marshal tensors into the C ABI of libfrosting_rasterizer.so
// (include/frosting_rasterizer.h -- frg_forward_ex / frg_backward / frg_mark_visible), on
// torch's CURRENT HIP stream under a device guard for means3D's device (the reference
// launches on the legacy default stream, e.g. forward.cu:389).
//
// Differences that a caller can observe, all permitted by the reference's contract:
//   * absent optional inputs are recognised by numel() == 0 (the reference tests
//     data_ptr() == nullptr, forward.cu:205,241);
//   * outputs are torch::empty, not zero-filled: every element is written by the kernels
//     (the reference must pre-zero ~300 B/Gaussian of gradients, rasterize_points.cu:151-159);
//   * scratch comes from torch's caching allocator through the C ABI's allocation callbacks
//     (replaces resizeFunctional, rasterize_points.cu:27-33): after the first view no
//     hipMalloc remains on the path, and -- unlike one shared arena -- several forwards may
//     be outstanding before their backwards run;
//   * one extra export, rasterize_gaussians_masked(..., keep_mask): Frosting's occlusion
//     culling as a per-Gaussian skip flag (frosting_scene/frosting_model.py:1564-1586).
#include <torch/extension.h>

#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <string>
#include <tuple>

#include "frosting_rasterizer.h"

namespace {

// One growable byte tensor per scratch chunk; `grow` is the frg_alloc_fn the C ABI calls.
struct Chunk {
    torch::Tensor t;
    torch::Device device;
    explicit Chunk(torch::Device d) : t(torch::empty({0}, torch::TensorOptions(torch::kByte).device(d))), device(d) {}
    static char* grow(void* user, size_t bytes)
    {
        Chunk* c = static_cast<Chunk*>(user);
        c->t = torch::empty({static_cast<int64_t>(bytes ? bytes : 1)}, torch::TensorOptions(torch::kByte).device(c->device));
        return reinterpret_cast<char*>(c->t.data_ptr());
    }
};

// The C ABI hands every callback the same `user`; three chunks -> three trampolines over one struct.
struct Chunks {
    Chunk geom, binning, img;
    explicit Chunks(torch::Device d) : geom(d), binning(d), img(d) {}
    static char* grow_geom(void* u, size_t n) { return Chunk::grow(&static_cast<Chunks*>(u)->geom, n); }
    static char* grow_binning(void* u, size_t n) { return Chunk::grow(&static_cast<Chunks*>(u)->binning, n); }
    static char* grow_img(void* u, size_t n) { return Chunk::grow(&static_cast<Chunks*>(u)->img, n); }
};

// float32 device pointer of an optional input; NULL when the tensor is the reference's
// "absent" encoding (empty tensor, DGR/diff_gaussian_rasterization/__init__.py:197-207)
const float* opt_f32(const torch::Tensor& t, const torch::Device& dev, const char* name, torch::Tensor& keep_alive)
{
    if (!t.defined() || t.numel() == 0) return nullptr;
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32, got ", t.scalar_type());
    TORCH_CHECK(t.device() == dev, name, " is on ", t.device(), ", expected ", dev);
    keep_alive = t.contiguous();
    return keep_alive.data_ptr<float>();
}

void check_rc(int rc, const char* what)
{
    TORCH_CHECK(rc >= 0, what, " failed (", rc, "): ", frg_last_error());
}

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
forward_common(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
               const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
               const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
               const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
               const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
               const bool prefiltered, const bool debug, const torch::Tensor* keep_mask, const bool forward_only = false,
               const int exact_blend = -1)
{
    TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");  // rasterize_points.cu:57-59
    TORCH_CHECK(means3D.is_cuda(), "frosting_amd rasterizer: means3D must live on a ROCm device (no CPU path)");
    const torch::Device dev = means3D.device();
    const c10::hip::HIPGuard guard(dev.index());
    const int P = static_cast<int>(means3D.size(0));
    const int H = image_height, W = image_width;

    torch::Tensor out_color = torch::empty({3, H, W}, means3D.options().dtype(torch::kFloat32));
    torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
    Chunks chunks(dev);

    torch::Tensor k[12];
    frg_forward_args a{};
    a.struct_size = sizeof(a);
    a.geometry_alloc = &Chunks::grow_geom;
    a.binning_alloc = &Chunks::grow_binning;
    a.image_alloc = &Chunks::grow_img;
    a.user = &chunks;
    a.P = P;
    a.D = degree;
    a.M = (sh.defined() && sh.numel() != 0) ? static_cast<int>(sh.size(1)) : 0;   // rasterize_points.cu:83-87
    a.background = opt_f32(background, dev, "background", k[0]);
    a.width = W;
    a.height = H;
    a.means3D = opt_f32(means3D, dev, "means3D", k[1]);
    a.shs = opt_f32(sh, dev, "sh", k[2]);
    a.colors_precomp = opt_f32(colors, dev, "colors", k[3]);
    a.opacities = opt_f32(opacity, dev, "opacity", k[4]);
    a.scales = opt_f32(scales, dev, "scales", k[5]);
    a.scale_modifier = scale_modifier;
    a.rotations = opt_f32(rotations, dev, "rotations", k[6]);
    a.cov3D_precomp = opt_f32(cov3D_precomp, dev, "cov3D_precomp", k[7]);
    a.viewmatrix = opt_f32(viewmatrix, dev, "viewmatrix", k[8]);
    a.projmatrix = opt_f32(projmatrix, dev, "projmatrix", k[9]);
    a.cam_pos = opt_f32(campos, dev, "campos", k[10]);
    a.tan_fovx = tan_fovx;
    a.tan_fovy = tan_fovy;
    a.prefiltered = prefiltered ? 1 : 0;
    a.out_color = out_color.data_ptr<float>();
    a.radii = P ? radii.data_ptr<int>() : nullptr;
    a.debug = debug ? 1 : 0;
    a.hip_stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    a.instance_capacity = 0;
    a.keep_mask = nullptr;
    a.forward_only = forward_only ? 1 : 0;
    TORCH_CHECK(exact_blend >= -1 && exact_blend <= 1, "exact_blend must be -1 (the process option), 0 or 1");
    a.exact_blend = exact_blend + 1;      // frg_forward_args: 0 = the process-wide option, k + 1 = value k for this call
    if (keep_mask && keep_mask->defined() && P && keep_mask->numel() != 0) {   // (the forward-only / _ex exports take an empty tensor for "no mask")
        TORCH_CHECK(keep_mask->dim() == 1 && keep_mask->size(0) == P && keep_mask->device() == dev &&
                        (keep_mask->scalar_type() == torch::kBool || keep_mask->scalar_type() == torch::kUInt8),
                    "keep_mask must be a bool / uint8 tensor of shape (num_points,) on the Gaussians' device");
        k[11] = keep_mask->contiguous();
        a.keep_mask = static_cast<const unsigned char*>(k[11].data_ptr());
    }
    const int rendered = frg_forward_ex(&a);
    check_rc(rendered, "frg_forward");
    return std::make_tuple(rendered, out_color, radii, chunks.geom.t, chunks.binning.t, chunks.img.t);
}

}  // namespace

// DGR/rasterize_points.h:18-38
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansHIP(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                      const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                      const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                      const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
                      const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                      const bool prefiltered, const bool debug)
{
    return forward_common(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                          projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                          nullptr);
}

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansMaskedHIP(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                            const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                            const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                            const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                            const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
                            const torch::Tensor& campos, const bool prefiltered, const bool debug,
                            const torch::Tensor& keep_mask)
{
    return forward_common(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                          projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                          &keep_mask);
}

// the forward of a call no backward will follow (frg_forward_args::forward_only): the 19 arguments + keep_mask (may be empty)
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansForwardOnlyHIP(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                                 const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                                 const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                                 const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                                 const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
                                 const torch::Tensor& campos, const bool prefiltered, const bool debug,
                                 const torch::Tensor& keep_mask)
{
    return forward_common(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                          projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                          &keep_mask, true);
}

// exact_blend: -1 = the arithmetic of the forward that filled the buffers (what the library remembers of it, else what it
// stamped into imageBuffer), 0 | 1 = the caller carried the forward's mode itself (the autograd ctx of frosting_amd/rasterizer.py)
static std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
backward_common(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug, const int exact_blend)
{
    TORCH_CHECK(exact_blend >= -1 && exact_blend <= 1, "exact_blend must be -1 (as the forward), 0 or 1");
    TORCH_CHECK(means3D.is_cuda(), "frosting_amd rasterizer: means3D must live on a ROCm device (no CPU path)");
    const torch::Device dev = means3D.device();
    const c10::hip::HIPGuard guard(dev.index());
    const int P = static_cast<int>(means3D.size(0));
    const int H = static_cast<int>(dL_dout_color.size(1)), W = static_cast<int>(dL_dout_color.size(2));  // rasterize_points.cu:142-143
    const bool has_sh = sh.defined() && sh.numel() != 0;
    const bool has_sr = scales.defined() && scales.numel() != 0;
    const int M = has_sh ? static_cast<int>(sh.size(1)) : 0;

    const auto f32 = means3D.options().dtype(torch::kFloat32);
    torch::Tensor dL_dmeans3D = torch::empty({P, 3}, f32), dL_dmeans2D = torch::empty({P, 3}, f32);
    torch::Tensor dL_dcolors = torch::empty({P, 3}, f32), dL_dopacity = torch::empty({P, 1}, f32);
    torch::Tensor dL_dcov3D = torch::empty({P, 6}, f32);
    // rows the kernels do not write (absent input) stay zero, as in the reference's zero-allocated outputs
    torch::Tensor dL_dsh = has_sh ? torch::empty({P, M, 3}, f32) : torch::zeros({P, M, 3}, f32);
    torch::Tensor dL_dscales = has_sr ? torch::empty({P, 3}, f32) : torch::zeros({P, 3}, f32);
    torch::Tensor dL_drotations = has_sr ? torch::empty({P, 4}, f32) : torch::zeros({P, 4}, f32);

    if (P != 0) {
        torch::Tensor k[11];
        const size_t ws_bytes = frg_backward_workspace_bytes(P, R);
        torch::Tensor workspace = torch::empty({static_cast<int64_t>(ws_bytes)}, torch::TensorOptions(torch::kByte).device(dev));
        torch::Tensor radii_c = radii.contiguous();
        TORCH_CHECK(radii_c.scalar_type() == torch::kInt32 && radii_c.device() == dev, "radii must be int32 on ", dev);
        TORCH_CHECK(geomBuffer.device() == dev && binningBuffer.device() == dev && imageBuffer.device() == dev,
                    "the forward's scratch buffers must be on ", dev);
        frg_backward_args a{};
        a.struct_size = sizeof(a);
        a.P = P; a.D = degree; a.M = M; a.R = R;
        a.background = opt_f32(background, dev, "background", k[0]);
        a.width = W; a.height = H;
        a.means3D = opt_f32(means3D, dev, "means3D", k[1]);
        a.shs = opt_f32(sh, dev, "sh", k[2]);
        a.colors_precomp = opt_f32(colors, dev, "colors", k[3]);
        a.scales = opt_f32(scales, dev, "scales", k[4]);
        a.scale_modifier = scale_modifier;
        a.rotations = opt_f32(rotations, dev, "rotations", k[5]);
        a.cov3D_precomp = opt_f32(cov3D_precomp, dev, "cov3D_precomp", k[6]);
        a.viewmatrix = opt_f32(viewmatrix, dev, "viewmatrix", k[7]);
        a.projmatrix = opt_f32(projmatrix, dev, "projmatrix", k[8]);
        a.campos = opt_f32(campos, dev, "campos", k[9]);
        a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
        a.radii = radii_c.data_ptr<int>();
        a.geom_buffer = reinterpret_cast<char*>(geomBuffer.data_ptr());
        a.binning_buffer = reinterpret_cast<char*>(binningBuffer.data_ptr());
        a.image_buffer = reinterpret_cast<char*>(imageBuffer.data_ptr());
        a.dL_dpix = opt_f32(dL_dout_color, dev, "dL_dout_color", k[10]);
        a.dL_dmean2D = dL_dmeans2D.data_ptr<float>();
        a.dL_dconic = nullptr;                      // never returned (rasterize_points.cu:195)
        a.dL_dopacity = dL_dopacity.data_ptr<float>();
        a.dL_dcolor = dL_dcolors.data_ptr<float>();
        a.dL_dmean3D = dL_dmeans3D.data_ptr<float>();
        a.dL_dcov3D = dL_dcov3D.data_ptr<float>();
        a.dL_dsh = has_sh ? dL_dsh.data_ptr<float>() : nullptr;
        a.dL_dscale = has_sr ? dL_dscales.data_ptr<float>() : nullptr;
        a.dL_drot = has_sr ? dL_drotations.data_ptr<float>() : nullptr;
        a.workspace = reinterpret_cast<char*>(workspace.data_ptr());
        a.workspace_bytes = ws_bytes;
        a.debug = debug ? 1 : 0;
        a.hip_stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        a.exact_blend = exact_blend + 1;            // frg_backward_args: 0 = as the forward, k + 1 = value k
        const int rc = frg_backward_ex(&a);
        check_rc(rc, "frg_backward");
        // `workspace` returns to the caching allocator here; reuse is ordered on this same stream
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

// DGR/rasterize_points.h:40-62; returns the reference's eight gradients in its order (rasterize_points.cu:195)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardHIP(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                              const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                              const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                              const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                              const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                              const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                              const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug)
{
    return backward_common(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                           projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer,
                           imageBuffer, debug, -1);
}

// The same with the forward's blend arithmetic handed over by the caller (who carried it beside the buffers -- the autograd
// ctx): nothing about the forward has to be remembered by the library or read back from the buffers.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardExHIP(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                                const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                                const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                                const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                                const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                                const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                                const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug,
                                const int exact_blend)
{
    return backward_common(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                           projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer,
                           imageBuffer, debug, exact_blend);
}

// every forward option of the Python layer in one export: the 19 arguments, keep_mask (may be empty), exact_blend
// (-1 = the process option | 0 | 1, for THIS call), forward_only
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansExHIP(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                        const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                        const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                        const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                        const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
                        const torch::Tensor& campos, const bool prefiltered, const bool debug,
                        const torch::Tensor& keep_mask, const int exact_blend, const bool forward_only)
{
    return forward_common(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                          projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                          &keep_mask, forward_only, exact_blend);
}

// DGR/rasterize_points.h:64-67, rasterize_points.cu:198-217
torch::Tensor markVisibleHIP(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix)
{
    TORCH_CHECK(means3D.is_cuda(), "frosting_amd rasterizer: means3D must live on a ROCm device (no CPU path)");
    const torch::Device dev = means3D.device();
    const c10::hip::HIPGuard guard(dev.index());
    const int P = static_cast<int>(means3D.size(0));
    torch::Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P != 0) {
        torch::Tensor k[3];
        const int rc = frg_mark_visible(P, opt_f32(means3D, dev, "means3D", k[0]), opt_f32(viewmatrix, dev, "viewmatrix", k[1]),
                                        opt_f32(projmatrix, dev, "projmatrix", k[2]),
                                        static_cast<unsigned char*>(present.data_ptr()),
                                        c10::hip::getCurrentHIPStream(dev.index()).stream());
        check_rc(rc, "frg_mark_visible");
    }
    return present;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "MI355X-native Gaussian-splat rasterizer (hand-written gfx950 HIP behind include/frosting_rasterizer.h)";
    m.def("rasterize_gaussians", &RasterizeGaussiansHIP);
    m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackwardHIP);
    m.def("mark_visible", &markVisibleHIP);
    m.def("rasterize_gaussians_masked", &RasterizeGaussiansMaskedHIP);
    m.def("rasterize_gaussians_forward_only", &RasterizeGaussiansForwardOnlyHIP);
    m.def("rasterize_gaussians_ex", &RasterizeGaussiansExHIP);
    m.def("rasterize_gaussians_backward_ex", &RasterizeGaussiansBackwardExHIP);
    m.def("get_option", [](const std::string& name) { return frg_get_option(name.c_str()); });
    m.def("library_version", []() { return frg_version(); });
}
