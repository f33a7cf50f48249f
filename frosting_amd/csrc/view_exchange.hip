// View-parallel gradient exchange helpers (no counterpart in the
// single-GPU reference).
//
// The SH gradient of one view is rank one per Gaussian (backward.cu:20-139):
//     dL_dsh_v[i][ch] = basis_i(normalize(mean - campos_v)) * dRGB_v[ch]
// where dRGB_v is the colour gradient after the clamp mask (backward.cu:31-34).  The
// basis depends only on replicated data (means3D, the camera centre), so the SUM over
// views of the 48-float SH gradient can be rebuilt on every rank from 3 floats per
// Gaussian and view.  The exchange therefore moves an all-gather of dRGB (3 floats)
// plus an all-reduce of the 11 other parameter gradients instead of an all-reduce of
// 59 floats per Gaussian: 2.6x fewer bytes over xGMI per step.
//
// The arithmetic of the basis and of the product is the very expression sequence of
// preprocess_bwd.hip (same contraction-off translation unit flags), so every term is
// bit-identical to the per-view dL_dsh and the result equals their sum taken in view order.
#include "gauss_math.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace frg {

#define VX_THREADS 256
#define VX_SUB 16
#define VX_ROW_F4 13

__device__ __forceinline__ void vx_wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// out[P,3] = dL_dcolor * (clamped ? 0 : 1); rows of culled Gaussians are zero in dL_dcolor already.
__global__ void __launch_bounds__(256)
sh_color_grad_kernel(int P, const float4* __restrict__ rgb_clamped, const int* __restrict__ radii,
                     const float* __restrict__ dL_dcolor, float* __restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    float r = 0.f, g = 0.f, b = 0.f;
    if (radii[idx] > 0) {
        const uint32_t bits = __float_as_uint(rgb_clamped[FRG_REC * idx].w);
        r = dL_dcolor[3 * idx] * ((bits & 1u) ? 0.f : 1.f);
        g = dL_dcolor[3 * idx + 1] * ((bits & 2u) ? 0.f : 1.f);
        b = dL_dcolor[3 * idx + 2] * ((bits & 4u) ? 0.f : 1.f);
    }
    out[3 * idx] = r; out[3 * idx + 1] = g; out[3 * idx + 2] = b;
}

// dL_dsh[P,M,3] = sum over views (in view order) of basis(dir_v) (x) dRGB_v.
// One lane per Gaussian for the basis; the 192-byte rows leave through a wave-private LDS
// transpose as contiguous float4 streams (same idiom as preprocess_bwd.hip).
template <bool SH16>
__global__ void __launch_bounds__(VX_THREADS)
sh_grad_from_views_kernel(int P, int D, int M, int n_views, const float* __restrict__ means3D,
                          const float* __restrict__ campos, long long campos_stride,
                          const float* __restrict__ drgb, long long view_stride, float* __restrict__ dL_dsh)
{
    __shared__ __attribute__((aligned(16))) float4 lds_all[(VX_THREADS / 64) * VX_SUB * VX_ROW_F4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* shbuf = lds_all + wave * VX_SUB * VX_ROW_F4;
    const int idx0 = (blockIdx.x * (VX_THREADS / 64) + wave) * 64;
    if (idx0 >= P) return;
    const int idx = idx0 + lane;
    const bool valid = idx < P;

    float out[48];
#pragma unroll
    for (int i = 0; i < 48; i++) out[i] = 0.0f;
    float3 mean = make_float3(0.f, 0.f, 0.f);
    if (valid) mean = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);

#pragma unroll 1
    for (int v = 0; v < n_views; v++) {
        if (!valid) continue;
        const float* g = drgb + (size_t)v * view_stride + 3 * (size_t)idx;
        const float dRGB[3] = {g[0], g[1], g[2]};
        if (dRGB[0] == 0.f && dRGB[1] == 0.f && dRGB[2] == 0.f) continue;   // culled or clamped in this view
        const float* cp = campos + (size_t)v * campos_stride;
        const float dox = mean.x - cp[0], doy = mean.y - cp[1], doz = mean.z - cp[2];
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);
        const float x = dox / len, y = doy / len, z = doz / len;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        float wgt[16];
#pragma unroll
        for (int i = 0; i < 16; i++) wgt[i] = 0.0f;
        wgt[0] = kSH0;
        if (D > 0) { wgt[1] = -kSH1 * y; wgt[2] = kSH1 * z; wgt[3] = -kSH1 * x; }
        if (D > 1) {
            wgt[4] = kSH2[0] * xy; wgt[5] = kSH2[1] * yz; wgt[6] = kSH2[2] * (2.f * zz - xx - yy);
            wgt[7] = kSH2[3] * xz; wgt[8] = kSH2[4] * (xx - yy);
        }
        if (D > 2) {
            wgt[9] = kSH3[0] * y * (3.f * xx - yy); wgt[10] = kSH3[1] * xy * z;
            wgt[11] = kSH3[2] * y * (4.f * zz - xx - yy); wgt[12] = kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
            wgt[13] = kSH3[4] * x * (4.f * zz - xx - yy); wgt[14] = kSH3[5] * z * (xx - yy);
            wgt[15] = kSH3[6] * x * (xx - 3.f * yy);
        }
#pragma unroll
        for (int i = 0; i < 48; i++) out[i] += wgt[i / 3] * dRGB[i % 3];
    }

    if (SH16) {
        float4* dst = reinterpret_cast<float4*>(dL_dsh) + (size_t)idx0 * 12;
        const int nvalid = min(64, P - idx0);
#pragma unroll 1
        for (int h = 0; h < 64 / VX_SUB; h++) {
            if ((lane / VX_SUB) == h) {
#pragma unroll
                for (int j = 0; j < 12; j++)
                    shbuf[(lane % VX_SUB) * VX_ROW_F4 + j] = make_float4(out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]);
            }
            vx_wave_fence();
#pragma unroll
            for (int k = 0; k < VX_SUB * 12 / 64; k++) {
                const int f = k * 64 + lane, gl = f / 12, j = f - gl * 12;
                if (h * VX_SUB + gl < nvalid) {     // (576 MB written once: non-temporal, like the per-view rows of preprocess_bwd.hip)
                    typedef float nt_f4 __attribute__((ext_vector_type(4)));
                    const float4 v = shbuf[gl * VX_ROW_F4 + j];
                    __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4*>(dst + (size_t)h * VX_SUB * 12 + f));
                }
            }
            vx_wave_fence();
        }
    } else if (valid) {
        float* o = dL_dsh + (size_t)idx * M * 3;
        const int n = min(M, 16) * 3;
#pragma unroll
        for (int i = 0; i < 48; i++)
            if (i < n) o[i] = out[i];
        for (int i = 48; i < M * 3; i++) o[i] = 0.0f;
    }
}

// ---- sparse exchange of the dense part (round 5) --------------------------------------------------------------------------
// Per view only the Gaussians some pixel reached before its tile saturated carry a gradient -- one visible Gaussian in
// seven at C3 -- so the rows that travel are (index, 11 dense floats, dRGB): 64 bytes per Gaussian WITH a gradient instead
// of 56 per Gaussian.  pack: every row that is not all zero, compacted (wave ballot + one atomic per wave: the order of the
// rows is whatever the waves' atomics made it -- indices are unique per view, so nothing downstream depends on it).
// scatter: one view's rows added into the dense gradient arrays (and its dRGB laid out densely for the SH rebuild above);
// one launch per view, in view order, on one stream: every element receives its terms in view order -- the sum is the
// single-process accumulation bit for bit.
#define FRG_ROW_FLOATS 16
#define PACK_ITEMS 8      // Gaussians per thread: one reservation (a device-scope atomic with return, ~10 ns each and serialised
                          // on the one counter) per 2048 Gaussians -- one per wave of 64 took 0.54 ms at 3 M Gaussians
__global__ void __launch_bounds__(256)
pack_grad_rows_kernel(int P, const float* __restrict__ g_means3D, const float* __restrict__ g_scales,
                      const float* __restrict__ g_rot, const float* __restrict__ g_opac, const float* __restrict__ drgb,
                      float4* __restrict__ rows, unsigned int capacity, unsigned int* __restrict__ count)
{
    __shared__ unsigned int wave_base[PACK_ITEMS][4], block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = blockIdx.x * (256 * PACK_ITEMS);
    // pass 1: which of this thread's Gaussians have a non-zero row (a NaN row is live: it must show up in the sum)
    uint64_t masks[PACK_ITEMS];
    unsigned int mine = 0;
#pragma unroll
    for (int it = 0; it < PACK_ITEMS; it++) {
        const int idx = first + it * 256 + (int)threadIdx.x;
        bool live = false;
        if (idx < P) {
            const float* m = g_means3D + 3 * (size_t)idx;
            const float* sc = g_scales + 3 * (size_t)idx;
            const float* q = g_rot + 4 * (size_t)idx;
            const float* d = drgb + 3 * (size_t)idx;
            live = (m[0] != 0.0f) | (m[1] != 0.0f) | (m[2] != 0.0f) | (sc[0] != 0.0f) | (sc[1] != 0.0f) | (sc[2] != 0.0f) |
                   (g_opac[idx] != 0.0f) | (q[0] != 0.0f) | (q[1] != 0.0f) | (q[2] != 0.0f) | (q[3] != 0.0f) |
                   (d[0] != 0.0f) | (d[1] != 0.0f) | (d[2] != 0.0f);
        }
        masks[it] = __builtin_amdgcn_ballot_w64(live);
        if (lane == 0) wave_base[it][wave] = (unsigned int)__popcll(masks[it]);
        mine |= live ? (1u << it) : 0u;
    }
    __syncthreads();
    // one reservation per block; the (item, wave) pieces get consecutive ranges in item-major order
    if (threadIdx.x == 0) {
        unsigned int run = 0;
#pragma unroll
        for (int it = 0; it < PACK_ITEMS; it++)
#pragma unroll
            for (int w = 0; w < 4; w++) { const unsigned int c = wave_base[it][w]; wave_base[it][w] = run; run += c; }
        block_base = run ? atomicAdd(count, run) : 0u;
    }
    __syncthreads();
    const unsigned int bb = block_base;
    // pass 2: the rows (the arrays are read again: they come from the L2)
#pragma unroll
    for (int it = 0; it < PACK_ITEMS; it++) {
        if (!((mine >> it) & 1u)) continue;
        const int idx = first + it * 256 + (int)threadIdx.x;
        const unsigned int at = bb + wave_base[it][wave] + (unsigned int)__popcll(masks[it] & ((1ull << lane) - 1ull));
        if (at >= capacity) continue;             // (over capacity: the count says so, the host packs again into a larger buffer)
        const float* m = g_means3D + 3 * (size_t)idx;
        const float* sc = g_scales + 3 * (size_t)idx;
        const float* q = g_rot + 4 * (size_t)idx;
        const float* d = drgb + 3 * (size_t)idx;
        float4* r = rows + (size_t)at * (FRG_ROW_FLOATS / 4);
        r[0] = make_float4(__uint_as_float((uint32_t)idx), m[0], m[1], m[2]);
        r[1] = make_float4(sc[0], sc[1], sc[2], g_opac[idx]);
        r[2] = make_float4(q[0], q[1], q[2], q[3]);
        r[3] = make_float4(d[0], d[1], d[2], 0.0f);
    }
}

__global__ void __launch_bounds__(256)
scatter_grad_rows_kernel(unsigned int n, int P, const float4* __restrict__ rows, float* __restrict__ g_means3D,
                         float* __restrict__ g_scales, float* __restrict__ g_rot, float* __restrict__ g_opac,
                         float* __restrict__ drgb_dense)
{
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4* r = rows + (size_t)i * (FRG_ROW_FLOATS / 4);
    const float4 a = r[0], b = r[1], c = r[2], d = r[3];
    const uint32_t idx = __float_as_uint(a.x);
    if (idx >= (uint32_t)P) return;              // (a corrupt row must not write out of range)
    float* m = g_means3D + 3 * (size_t)idx;
    m[0] += a.y; m[1] += a.z; m[2] += a.w;
    float* sc = g_scales + 3 * (size_t)idx;
    sc[0] += b.x; sc[1] += b.y; sc[2] += b.z;
    g_opac[idx] += b.w;
    float* q = g_rot + 4 * (size_t)idx;
    q[0] += c.x; q[1] += c.y; q[2] += c.z; q[3] += c.w;
    if (drgb_dense) { float* o = drgb_dense + 3 * (size_t)idx; o[0] = d.x; o[1] = d.y; o[2] = d.z; }
}

hipError_t launch_pack_grad_rows(int P, const float* g_means3D, const float* g_scales, const float* g_rot, const float* g_opac,
                                 const float* drgb, float* rows, unsigned int capacity, unsigned int* count, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pack_grad_rows_kernel, dim3((P + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS)), dim3(256), 0, s, P, g_means3D, g_scales, g_rot, g_opac, drgb,
                       reinterpret_cast<float4*>(rows), capacity, count);
    return hipGetLastError();
}

hipError_t launch_scatter_grad_rows(unsigned int n, int P, const float* rows, float* g_means3D, float* g_scales, float* g_rot,
                                    float* g_opac, float* drgb_dense, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_grad_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, P, reinterpret_cast<const float4*>(rows),
                       g_means3D, g_scales, g_rot, g_opac, drgb_dense);
    return hipGetLastError();
}

hipError_t launch_sh_color_grad(int P, const GeomState& g, const int* radii, const float* dL_dcolor, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(sh_color_grad_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g.rgb_clamped, radii, dL_dcolor, out);
    return hipGetLastError();
}

hipError_t launch_sh_grad_from_views(int P, int D, int M, int n_views, const float* means3D, const float* campos,
                                     long long campos_stride, const float* drgb, long long view_stride, float* dL_dsh,
                                     hipStream_t s)
{
    const dim3 grid((P + VX_THREADS - 1) / VX_THREADS), block(VX_THREADS);
    const bool sh16 = M == 16 && (reinterpret_cast<uintptr_t>(dL_dsh) % 16 == 0);
    if (sh16)
        hipLaunchKernelGGL(sh_grad_from_views_kernel<true>, grid, block, 0, s, P, D, M, n_views, means3D, campos,
                           campos_stride, drgb, view_stride, dL_dsh);
    else
        hipLaunchKernelGGL(sh_grad_from_views_kernel<false>, grid, block, 0, s, P, D, M, n_views, means3D, campos,
                           campos_stride, drgb, view_stride, dL_dsh);
    return hipGetLastError();
}

}  // namespace frg
