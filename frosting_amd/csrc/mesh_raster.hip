// Triangle occlusion raster: the z-buffered "which face does each pixel see" pass that
// Frosting runs on the shell base mesh for occlusion culling and texture baking
// (frosting_utils/mesh_rasterization.py:109-156 -> frosting_utils/nvdiffrast.py:53,
// third-party nvdiffrast `dr.rasterize`, not in the reference tree).  Same contract:
//   pos  [V,4] clip-space vertices (x, y, z, w) = [v,1] @ full_proj_transform
//   tri  [F,3] int32
//   rast [H,W,4] float32 = (u, v, z/w, triangle_id + 1), all zero where no triangle;
//        u, v = perspective-correct barycentrics of vertex 0 and 1; pixel (col i, row j) is
//        sampled at NDC ((i+0.5)/W*2-1, (j+0.5)/H*2-1); fragments outside -1 <= z/w <= 1 or
//        behind the eye are discarded; nearest z/w wins, ties go to the smaller triangle id;
//        coverage by the top-left fill rule on exactly shared edge functions (see FILL RULE below).
//
// 2-D homogeneous rasterisation (edge functions from the adjugate of [x y w]), so triangles
// that cross the w = 0 plane need no clipping.  One thread per triangle walks its pixel
// bounding box and depth-tests with a 64-bit atomicMin on (ordered depth << 32 | id);
// triangles covering many pixels are deferred to a workgroup-per-triangle pass.  A resolve
// pass recomputes the attributes of each pixel's winner.
#include "../../include/frosting_rasterizer.h"
#include "frg_common.h"
#include <algorithm>

namespace frg {

#define MESH_BIG_AREA 1024  // bounding boxes above this many pixels go to the cooperative pass

struct TriSetup {
    // Edge functions e_i(X,Y) = a_i X + b_i Y + c_i over NDC, e_i = sign(det) * det[(X,Y,1), v_{i+1}, v_{i+2}]
    // (2-D homogeneous rasterisation): the pixel is covered iff every e_i >= 0 under the tie rule below, and
    // e_i / sum(e) is the perspective-correct barycentric of vertex i.  Float64, evaluated with the same
    // fused expression from both sides of a shared edge -- see edge_coeffs().
    double a[3], b[3], c[3];
    float z[3], w[3];
    int x0, y0, x1, y1;  // pixel bounding box [x0,x1) x [y0,y1)
    bool ok;
};

__device__ __forceinline__ uint32_t ordered_depth(float z)
{
    const uint32_t u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint
}

// FILL RULE.  Coverage is decided by the three edge functions at the pixel centre
// ((i + 0.5) / W * 2 - 1, (j + 0.5) / H * 2 - 1) -- OpenGL's sample position, which is what nvdiffrast's
// rasterizers use.  A centre strictly inside (all e > 0) is covered; a centre exactly ON an edge (e == 0)
// belongs to the triangle for which that edge is a "left" edge (a > 0), or a horizontal one with b > 0: the
// top-left rule of D3D / the usual OpenGL implementations, written on the edge normal (a, b).  Two
// triangles sharing an edge see it with opposite normals, so exactly one of them owns the centre: no
// double hits, no cracks.
// For that to hold in floating point both triangles must compute the SAME value for the shared edge.
// The coefficients are therefore built from the edge's two end points in a canonical order (by position
// bits, so duplicated vertices -- sphere poles, UV seams -- behave like shared ones) and only negated,
// which is exact, when the triangle traverses the edge the other way or faces away; the evaluation
// fma(a, X, fma(b, Y, c)) commutes with that negation bit for bit.
__device__ __forceinline__ bool pos_less(const float4 p, const float4 q)
{
    if (p.x != q.x) return p.x < q.x;
    if (p.y != q.y) return p.y < q.y;
    return p.w < q.w;
}

__device__ __forceinline__ void edge_coeffs(const float4 p, const float4 q, double sgn, double& a, double& b, double& c)
{
    // det[(X,Y,1), p, q] = X (p.y q.w - q.y p.w) + Y (q.x p.w - p.x q.w) + (p.x q.y - q.x p.y)
    const bool swap = pos_less(q, p);
    const float4 lo = swap ? q : p, hi = swap ? p : q;
    const double lx = lo.x, ly = lo.y, lw = lo.w, hx = hi.x, hy = hi.y, hw = hi.w;
    const double ca = fma(ly, hw, -(hy * lw)), cb = fma(hx, lw, -(lx * hw)), cc = fma(lx, hy, -(hx * ly));
    const double s = swap ? -sgn : sgn;
    a = s * ca; b = s * cb; c = s * cc;
}

__device__ __forceinline__ TriSetup tri_setup(const float4 v0, const float4 v1, const float4 v2, int W, int H)
{
    TriSetup t;
    t.ok = false;
    // bounding box: exact when every vertex is in front of the eye, whole screen otherwise
    if (v0.w > 1e-6f && v1.w > 1e-6f && v2.w > 1e-6f) {
        const float nx0 = v0.x / v0.w, nx1 = v1.x / v1.w, nx2 = v2.x / v2.w;
        const float ny0 = v0.y / v0.w, ny1 = v1.y / v1.w, ny2 = v2.y / v2.w;
        const float mnx = fminf(nx0, fminf(nx1, nx2)), mxx = fmaxf(nx0, fmaxf(nx1, nx2));
        const float mny = fminf(ny0, fminf(ny1, ny2)), mxy = fmaxf(ny0, fmaxf(ny1, ny2));
        if (mxx < -1.f || mnx > 1.f || mxy < -1.f || mny > 1.f) return t;
        // pixel i centre at ((i + 0.5) / W) * 2 - 1  =>  i = (ndc + 1) * W / 2 - 0.5.  A covered centre lies inside the
        // vertices' NDC extent: i in [imin, imax].  These float expressions are within ~1e-3 pixel of the true bounds (three
        // roundings of values below 2^13), so 1/16 pixel of slack keeps the box a superset -- coverage itself is decided
        // in double by tri_sample.  (Rounds 1-4: a whole pixel of slack on each side and two past the end -- a 2 x 2 pixel
        // triangle of the C4 shell sampled 36 centres instead of 9.)
        t.x0 = max(0, (int)floorf((mnx + 1.f) * 0.5f * W - 0.5f - 0.0625f));
        t.x1 = min(W, (int)floorf((mxx + 1.f) * 0.5f * W - 0.5f + 0.0625f) + 1);
        t.y0 = max(0, (int)floorf((mny + 1.f) * 0.5f * H - 0.5f - 0.0625f));
        t.y1 = min(H, (int)floorf((mxy + 1.f) * 0.5f * H - 0.5f + 0.0625f) + 1);
    } else if (v0.w <= 1e-6f && v1.w <= 1e-6f && v2.w <= 1e-6f) {
        return t;  // entirely behind the eye
    } else {
        t.x0 = 0; t.y0 = 0; t.x1 = W; t.y1 = H;
    }
    // (a box without a pixel centre -- sub-pixel triangles between centres, at the poles and the limb -- needs no setup)
    if (!(t.x1 > t.x0 && t.y1 > t.y0)) return t;
    // det of M = [[x0 x1 x2],[y0 y1 y2],[w0 w1 w2]] (double: it cancels badly for slivers); its sign is the facing
    const double x0 = v0.x, y0 = v0.y, w0 = v0.w, x1 = v1.x, y1 = v1.y, w1 = v1.w, x2 = v2.x, y2 = v2.y, w2 = v2.w;
    const double det = x0 * (y1 * w2 - y2 * w1) + y0 * (x2 * w1 - x1 * w2) + w0 * (x1 * y2 - x2 * y1);
    if (!(det != 0.0) || det != det) return t;   // degenerate (zero area in homogeneous space)
    const double sgn = det > 0.0 ? 1.0 : -1.0;    // both facings are rasterised (nvdiffrast does not cull)
    edge_coeffs(v1, v2, sgn, t.a[0], t.b[0], t.c[0]);
    edge_coeffs(v2, v0, sgn, t.a[1], t.b[1], t.c[1]);
    edge_coeffs(v0, v1, sgn, t.a[2], t.b[2], t.c[2]);
    t.z[0] = v0.z; t.z[1] = v1.z; t.z[2] = v2.z;
    t.w[0] = v0.w; t.w[1] = v1.w; t.w[2] = v2.w;
    t.ok = true;
    return t;
}

__device__ __forceinline__ bool edge_owns(double e, double a, double b)
{
    return e > 0.0 || (e == 0.0 && (a > 0.0 || (a == 0.0 && b > 0.0)));
}

// Evaluate one pixel; returns true and (u, v, z/w) when the pixel centre is covered.
__device__ __forceinline__ bool tri_sample(const TriSetup& t, int px, int py, int W, int H, float& u, float& v, float& zw)
{
    const double X = ((double)px + 0.5) / (double)W * 2.0 - 1.0;
    const double Y = ((double)py + 0.5) / (double)H * 2.0 - 1.0;
    const double e0 = fma(t.a[0], X, fma(t.b[0], Y, t.c[0]));
    const double e1 = fma(t.a[1], X, fma(t.b[1], Y, t.c[1]));
    const double e2 = fma(t.a[2], X, fma(t.b[2], Y, t.c[2]));
    if (!edge_owns(e0, t.a[0], t.b[0]) || !edge_owns(e1, t.a[1], t.b[1]) || !edge_owns(e2, t.a[2], t.b[2])) return false;
    const double s = e0 + e1 + e2;         // = |det| / w at the pixel
    if (!(s > 0.0)) return false;          // exactly degenerate
    const double r = 1.0 / s;
    const double b0 = e0 * r, b1 = e1 * r, b2 = e2 * r;
    const double zc = b0 * t.z[0] + b1 * t.z[1] + b2 * t.z[2];
    const double wc = b0 * t.w[0] + b1 * t.w[1] + b2 * t.w[2];
    zw = (float)(zc / wc);
    if (!(zw >= -1.f && zw <= 1.f)) return false;
    u = (float)b0; v = (float)b1;
    return true;
}

__device__ __forceinline__ TriSetup load_tri(int f, const float4* __restrict__ pos, const int* __restrict__ tri, int V, int W, int H)
{
    const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    if ((unsigned)i0 >= (unsigned)V || (unsigned)i1 >= (unsigned)V || (unsigned)i2 >= (unsigned)V) {
        TriSetup t; t.ok = false; return t;
    }
    return tri_setup(pos[i0], pos[i1], pos[i2], W, H);
}

__global__ void __launch_bounds__(256)
mesh_clear_kernel(size_t n, unsigned long long* __restrict__ depth, uint32_t* __restrict__ big_count)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) depth[i] = ~0ull;
    if (i == 0) *big_count = 0;
}

__global__ void __launch_bounds__(256)
mesh_raster_small_kernel(int V, int F, const float4* __restrict__ pos, const int* __restrict__ tri, int W, int H,
                         unsigned long long* __restrict__ depth, uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const TriSetup t = load_tri(f, pos, tri, V, W, H);
    if (!t.ok) return;
    if ((long long)(t.x1 - t.x0) * (t.y1 - t.y0) > MESH_BIG_AREA) {
        big_list[atomicAdd(big_count, 1u)] = (uint32_t)f;
        return;
    }
    for (int py = t.y0; py < t.y1; py++)
        for (int px = t.x0; px < t.x1; px++) {
            float u, v, zw;
            if (tri_sample(t, px, py, W, H, u, v, zw))
                atomicMin(&depth[(size_t)py * W + px], ((unsigned long long)ordered_depth(zw) << 32) | (uint32_t)f);
        }
}

__global__ void __launch_bounds__(256)
mesh_raster_big_kernel(int V, const float4* __restrict__ pos, const int* __restrict__ tri, int W, int H,
                       unsigned long long* __restrict__ depth, const uint32_t* __restrict__ big_list,
                       const uint32_t* __restrict__ big_count)
{
    const uint32_t n = *big_count;
    for (uint32_t b = blockIdx.x; b < n; b += gridDim.x) {
        const int f = (int)big_list[b];
        const TriSetup t = load_tri(f, pos, tri, V, W, H);
        if (!t.ok) continue;
        const int bw = t.x1 - t.x0;
        const long long area = (long long)bw * (t.y1 - t.y0);
        for (long long p = threadIdx.x; p < area; p += 256) {
            const int px = t.x0 + (int)(p % bw), py = t.y0 + (int)(p / bw);
            float u, v, zw;
            if (tri_sample(t, px, py, W, H, u, v, zw))
                atomicMin(&depth[(size_t)py * W + px], ((unsigned long long)ordered_depth(zw) << 32) | (uint32_t)f);
        }
    }
}

__global__ void __launch_bounds__(256)
mesh_resolve_kernel(int V, const float4* __restrict__ pos, const int* __restrict__ tri, int W, int H,
                    const unsigned long long* __restrict__ depth, float4* __restrict__ rast)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const unsigned long long key = depth[i];
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (key != ~0ull) {
        const int f = (int)(uint32_t)key;
        const TriSetup t = load_tri(f, pos, tri, V, W, H);
        float u = 0.f, v = 0.f, zw = 0.f;
        tri_sample(t, (int)(i % W), (int)(i / W), W, H, u, v, zw);
        out = make_float4(u, v, zw, (float)(f + 1));
    }
    rast[i] = out;
}

// visible-face form: no attributes, only "is this face the nearest surface of some pixel"
__global__ void __launch_bounds__(256)
mesh_mark_kernel(size_t n, const unsigned long long* __restrict__ depth, unsigned char* __restrict__ face_visible)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = depth[i];
    if (key != ~0ull) face_visible[(uint32_t)key] = 1;      // (several pixels store the same byte: benign)
}

// frg_mesh_occlusion_mask: everything the z-buffer pass needs, in one launch -- the clip-space vertices [v,1] @ M (the
// caller's torch.ones + torch.cat + matmul: frosting_utils/nvdiffrast.py:44-50), the cleared depth plane, the cleared face
// marks and the cleared list of big triangles.  A fused multiply-add per term in the order of the sum over k, as a GEMM's
// inner loop accumulates them.
__global__ void __launch_bounds__(256)
mesh_prepare_kernel(size_t n, unsigned long long* __restrict__ depth, uint32_t* __restrict__ big_count, int V,
                    const float* __restrict__ verts, const float* __restrict__ M, float4* __restrict__ pos, int F,
                    unsigned char* __restrict__ face_visible)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) depth[i] = ~0ull;
    if (i == 0) *big_count = 0;
    if (i < (size_t)F) face_visible[i] = 0;
    if (i < (size_t)V) {
        const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = fmaf(1.0f, M[12 + j], fmaf(z, M[8 + j], fmaf(y, M[4 + j], x * M[j])));
        pos[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// keep[i] = face_visible[cell_of_point[i]] for the shell's Gaussians, 1 for the background ones behind them
// (frosting_model.py:1564-1586: _index_mask[self._point_cell_indices] ++ ones)
#define MESH_KEEP_PER_THREAD 8
__global__ void __launch_bounds__(256)
mesh_keep_kernel(int n_shell, const long long* __restrict__ cell_of_point, int F, const unsigned char* __restrict__ face_visible,
                 int n_total, unsigned char* __restrict__ keep)
{
    // eight consecutive Gaussians per thread: the indices as four 16-byte loads, the marks from the (cache-resident) face
    // table, the flags as one 8-byte store where the block is whole and both arrays are aligned for it
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * MESH_KEEP_PER_THREAD;
    if (i0 >= n_total) return;
    auto one = [&](long long i) -> unsigned char {
        if (i >= n_shell) return 1;
        long long c = cell_of_point[i];
        if (c < 0) c += F;                                   // (torch indexing: negative indices count from the end)
        return (c >= 0 && c < F) ? face_visible[c] : 0;
    };
    const bool whole = i0 + MESH_KEEP_PER_THREAD <= n_shell && (reinterpret_cast<uintptr_t>(cell_of_point) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(keep) & 7) == 0;
    if (whole) {
        long long c[MESH_KEEP_PER_THREAD];
#pragma unroll
        for (int k = 0; k < MESH_KEEP_PER_THREAD; k += 2) {
            const longlong2 v = *reinterpret_cast<const longlong2*>(cell_of_point + i0 + k);
            c[k] = v.x; c[k + 1] = v.y;
        }
        unsigned long long out = 0;
#pragma unroll
        for (int k = 0; k < MESH_KEEP_PER_THREAD; k++) {
            long long ck = c[k] < 0 ? c[k] + F : c[k];
            const unsigned char m = (ck >= 0 && ck < F) ? face_visible[ck] : 0;
            out |= (unsigned long long)m << (8 * k);
        }
        *reinterpret_cast<unsigned long long*>(keep + i0) = out;
        return;
    }
    for (long long i = i0; i < i0 + MESH_KEEP_PER_THREAD && i < n_total; i++) keep[i] = one(i);
}

// the z-buffer pass shared by both entry points
static void launch_depth_pass(int V, int F, const float4* p4, const int* tri, int width, int height, char* workspace, hipStream_t s,
                              unsigned long long*& depth, bool prepared = false)
{
    const size_t N = (size_t)width * height;
    depth = reinterpret_cast<unsigned long long*>(workspace);
    uint32_t* big_list = reinterpret_cast<uint32_t*>(workspace + align_up(N * 8, 256));
    uint32_t* big_count = reinterpret_cast<uint32_t*>(workspace + align_up(N * 8, 256) + align_up((size_t)(F > 0 ? F : 1) * 4, 256));
    if (!prepared) hipLaunchKernelGGL(mesh_clear_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, depth, big_count);
    if (F > 0) {
        hipLaunchKernelGGL(mesh_raster_small_kernel, dim3((F + 255) / 256), dim3(256), 0, s, V, F, p4, tri, width, height,
                           depth, big_list, big_count);
        // (usually there is no such triangle: a grid the GPU holds at once, the workgroups stride)
        hipLaunchKernelGGL(mesh_raster_big_kernel, dim3(1024), dim3(256), 0, s, V, p4, tri, width, height, depth,
                           big_list, big_count);
    }
}

}  // namespace frg

extern "C" {

int frg_mesh_visible_faces(int V, int F, const float* pos, const int* tri, int width, int height, unsigned char* face_visible,
                           char* workspace, size_t workspace_bytes, void* hip_stream)
{
    if (V < 0 || F < 0 || width <= 0 || height <= 0) return FRG_EINVAL;
    if (F == 0) return FRG_OK;
    if (!pos || !tri || !face_visible) return FRG_EINVAL;
    if (!workspace || workspace_bytes < frg_mesh_raster_workspace_bytes(F, width, height)) return FRG_EALLOC;
    hipStream_t s = (hipStream_t)hip_stream;
    if (hipMemsetAsync(face_visible, 0, (size_t)F, s) != hipSuccess) return FRG_EHIP;
    unsigned long long* depth = nullptr;
    frg::launch_depth_pass(V, F, reinterpret_cast<const float4*>(pos), tri, width, height, workspace, s, depth);
    const size_t N = (size_t)width * height;
    hipLaunchKernelGGL(frg::mesh_mark_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, depth, face_visible);
    return hipGetLastError() == hipSuccess ? FRG_OK : FRG_EHIP;
}

size_t frg_mesh_occlusion_workspace_bytes(int V, int F, int width, int height)
{
    return frg_mesh_raster_workspace_bytes(F, width, height) + frg::align_up((size_t)(V > 0 ? V : 1) * 16, 256);
}

int frg_mesh_occlusion_mask(int V, int F, const float* verts, const float* full_proj_transform, const int* tri, int width, int height,
                            int n_shell, const long long* cell_of_point, int n_background, unsigned char* keep,
                            unsigned char* face_visible, char* workspace, size_t workspace_bytes, void* hip_stream)
{
    if (V < 0 || F < 0 || width <= 0 || height <= 0 || n_shell < 0 || n_background < 0) return FRG_EINVAL;
    if ((long long)n_shell + n_background > 0x7fffffffLL) return FRG_EINVAL;
    if ((F > 0 && (!verts || !full_proj_transform || !tri || !face_visible)) || (n_shell > 0 && !cell_of_point) ||
        (n_shell + n_background > 0 && !keep))
        return FRG_EINVAL;
    if (!workspace || workspace_bytes < frg_mesh_occlusion_workspace_bytes(V, F, width, height)) return FRG_EALLOC;
    hipStream_t s = (hipStream_t)hip_stream;
    const size_t N = (size_t)width * height;
    const size_t raster_bytes = frg_mesh_raster_workspace_bytes(F, width, height);
    float4* p4 = reinterpret_cast<float4*>(workspace + raster_bytes);
    if (F > 0) {
        unsigned long long* depth = reinterpret_cast<unsigned long long*>(workspace);
        uint32_t* big_count = reinterpret_cast<uint32_t*>(workspace + frg::align_up(N * 8, 256) + frg::align_up((size_t)F * 4, 256));
        const size_t items = std::max(N, std::max((size_t)V, (size_t)F));
        hipLaunchKernelGGL(frg::mesh_prepare_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, N, depth, big_count, V, verts,
                           full_proj_transform, p4, F, face_visible);
        frg::launch_depth_pass(V, F, p4, tri, width, height, workspace, s, depth, true);
        hipLaunchKernelGGL(frg::mesh_mark_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, depth, face_visible);
    }
    const int n_total = n_shell + n_background;
    if (n_total > 0)
        hipLaunchKernelGGL(frg::mesh_keep_kernel, dim3((unsigned)((n_total + 256 * MESH_KEEP_PER_THREAD - 1) / (256 * MESH_KEEP_PER_THREAD))), dim3(256), 0, s, n_shell, cell_of_point, F,
                           face_visible, n_total, keep);
    return hipGetLastError() == hipSuccess ? FRG_OK : FRG_EHIP;
}

size_t frg_mesh_raster_workspace_bytes(int F, int width, int height)
{
    return frg::align_up((size_t)width * height * 8, 256) + frg::align_up((size_t)(F > 0 ? F : 1) * 4, 256) + 256;
}

int frg_mesh_rasterize(int V, int F, const float* pos, const int* tri, int width, int height, float* rast,
                       char* workspace, size_t workspace_bytes, void* hip_stream)
{
    if (V < 0 || F < 0 || width <= 0 || height <= 0 || !rast) return FRG_EINVAL;
    if (!workspace || workspace_bytes < frg_mesh_raster_workspace_bytes(F, width, height)) return FRG_EALLOC;
    if (F > 0 && (!pos || !tri)) return FRG_EINVAL;
    hipStream_t s = (hipStream_t)hip_stream;
    const size_t N = (size_t)width * height;
    const float4* p4 = reinterpret_cast<const float4*>(pos);
    unsigned long long* depth = nullptr;
    frg::launch_depth_pass(V, F, p4, tri, width, height, workspace, s, depth);
    hipLaunchKernelGGL(frg::mesh_resolve_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, V, p4, tri, width, height,
                       depth, reinterpret_cast<float4*>(rast));
    return hipGetLastError() == hipSuccess ? FRG_OK : FRG_EHIP;
}

}  // extern "C"
