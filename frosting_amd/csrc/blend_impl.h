// Front-to-back alpha compositing, forward and backward (the reference's K6 / K7:
// forward.cu:261-374, backward.cu:399-557), re-designed for CDNA4 wave64:
//
//   * one WAVE per 16x16 tile, each lane owns 4 pixels (one per 8x8 quadrant), so a
//     tile needs no workgroup barrier and no cross-wave reduction, and every LDS
//     broadcast read of a staged Gaussian is amortised over 4 pixel evaluations;
//   * instances are staged 64 at a time through LDS from 16-byte per-Gaussian
//     records (three global_load_dwordx4 gathers per instance);
//   * backward: the per-(tile,Gaussian) gradient is reduced across the wave with a
//     fixed DPP tree and written ONCE, without atomics, into the instance's
//     Gaussian-major slot; the per-Gaussian backward kernel sums the slots in a
//     fixed order.  The reference issues 9 global float atomics per (pixel,
//     Gaussian) pair (backward.cu:523,545-554) and is not reproducible run to run.
//
// EXACT=true : IEEE operation order of the reference, no contraction, accurate expf
//              (bit-identical image to the reference built with -ffp-contract=off).
// EXACT=false: FMA contraction and native exp2 (default product path).
// (template bodies; instantiated by blend_exact.hip / blend_fast.hip, which fix the
// floating-point contraction mode for everything below)
#pragma once
#include "frg_common.h"

namespace frg {

template <bool EXACT>
struct BlendMath;

template <>
struct BlendMath<true> {
    static __device__ __forceinline__ float power(float2 xy, float4 co, float px, float py, float& dx, float& dy)
    {
        dx = xy.x - px; dy = xy.y - py;
        return -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
    }
    static __device__ __forceinline__ float expo(float p) { return expf(p); }
    static __device__ __forceinline__ float mul3(float a, float b, float c) { return a * b * c; }
};

template <>
struct BlendMath<false> {
    static __device__ __forceinline__ float power(float2 xy, float4 co, float px, float py, float& dx, float& dy)
    {
        dx = xy.x - px; dy = xy.y - py;
        return -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
    }
    static __device__ __forceinline__ float expo(float p) { return __expf(p); }
    static __device__ __forceinline__ float mul3(float a, float b, float c) { return a * b * c; }
};

// lane -> pixel of quadrant q inside the tile
__device__ __forceinline__ void lane_pixel(int lane, int q, int tx, int ty, int& px, int& py)
{
    px = tx * FRG_TILE + (q & 1) * 8 + (lane & 7);
    py = ty * FRG_TILE + (q >> 1) * 8 + (lane >> 3);
}

// ---------------------------------------------------------------------------
template <bool EXACT>
__global__ void __launch_bounds__(64)
blend_fwd_kernel(int T, int gx, int W, int H, const uint2* __restrict__ ranges,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ xydr,
                 const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb_clamped,
                 const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                 float* __restrict__ out_color)
{
    using M = BlendMath<EXACT>;
    const int tile = xcd_tile_of_block(blockIdx.x, T);
    if (tile < 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);

    __shared__ float2 s_xy[64];
    __shared__ float4 s_co[64];
    __shared__ float4 s_rgb[64];

    float pxf[4], pyf[4], Tr[4], C[4][3];
    uint32_t last[4];
    bool inside[4];
    uint32_t live = 0;  // bit q set while pixel q still blends
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int px, py;
        lane_pixel(lane, q, tx, ty, px, py);
        pxf[q] = (float)px; pyf[q] = (float)py;
        inside[q] = px < W && py < H;
        if (inside[q]) live |= 1u << q;
        Tr[q] = 1.0f; C[q][0] = C[q][1] = C[q][2] = 0.0f; last[q] = 0;
    }

    for (int base = 0; base < n; base += 64) {
        if (__ballot(live != 0) == 0ull) break;  // whole tile saturated
        const int cnt = min(64, n - base);
        __syncthreads();
        if (lane < cnt) {
            const uint32_t id = point_list[rg.x + base + lane];
            const float4 a = xydr[id];
            s_xy[lane] = make_float2(a.x, a.y);
            s_co[lane] = conic_opacity[id];
            s_rgb[lane] = rgb_clamped[id];
        }
        __syncthreads();
        for (int j = 0; live != 0 && j < cnt; j++) {
            const float2 xy = s_xy[j];
            const float4 co = s_co[j];
            const uint32_t contributor = (uint32_t)(base + j + 1);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!(live & (1u << q))) continue;
                float dx, dy;
                const float power = M::power(xy, co, pxf[q], pyf[q], dx, dy);
                if (power > 0.0f) continue;
                const float alpha = fminf(0.99f, co.w * M::expo(power));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = Tr[q] * (1 - alpha);
                if (test_T < 0.0001f) { live &= ~(1u << q); continue; }
                const float4 col = s_rgb[j];
                C[q][0] += M::mul3(col.x, alpha, Tr[q]);
                C[q][1] += M::mul3(col.y, alpha, Tr[q]);
                C[q][2] += M::mul3(col.z, alpha, Tr[q]);
                Tr[q] = test_T;
                last[q] = contributor;
            }
        }
    }

    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (!inside[q]) continue;
        const size_t pid = (size_t)pyf[q] * W + (size_t)pxf[q];
        final_T[pid] = Tr[q];
        n_contrib[pid] = last[q];
        out_color[pid] = C[q][0] + Tr[q] * bg0;
        out_color[plane + pid] = C[q][1] + Tr[q] * bg1;
        out_color[2 * plane + pid] = C[q][2] + Tr[q] * bg2;
    }
}

// ---------------------------------------------------------------------------
// wave64 sum with a fixed DPP tree; the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(t);
}

__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_step<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_step<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_step<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_step<0x140, 0xf>(v);  // row_mirror         -> every lane holds its row's sum
    v = dpp_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the total
    return v;
}

template <bool EXACT>
__global__ void __launch_bounds__(64)
blend_bwd_kernel(int T, int gx, int gy, int W, int H, const uint2* __restrict__ ranges,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ xydr,
                 const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb_clamped,
                 const uint32_t* __restrict__ point_offsets, const float* __restrict__ bg,
                 const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                 const float* __restrict__ dL_dpix, float* __restrict__ slots, uint2* __restrict__ cutoff)
{
    using M = BlendMath<EXACT>;
    const int tile = xcd_tile_of_block(blockIdx.x, T);
    if (tile < 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const uint2 rg = ranges[tile];

    __shared__ float2 s_xy[64];
    __shared__ float4 s_co[64];
    __shared__ float4 s_rgb[64];
    __shared__ float s_part[64 * FRG_SLOT_FLOATS];

    float pxf[4], pyf[4], Tr[4], Tfin[4], dLp[4][3], accum[4][3], lastc[4][3], lasta[4], bgdot[4];
    uint32_t lastcon[4];
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const size_t plane = (size_t)H * W;
    uint32_t maxc = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int px, py;
        lane_pixel(lane, q, tx, ty, px, py);
        pxf[q] = (float)px; pyf[q] = (float)py;
        const bool inside = px < W && py < H;
        const size_t pid = (size_t)py * W + px;
        Tfin[q] = inside ? final_T[pid] : 0.0f;
        Tr[q] = Tfin[q];
        lastcon[q] = inside ? n_contrib[pid] : 0u;
        maxc = max(maxc, lastcon[q]);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            dLp[q][ch] = inside ? dL_dpix[ch * plane + pid] : 0.0f;
            accum[q][ch] = 0.0f; lastc[q][ch] = 0.0f;
        }
        lasta[q] = 0.0f;
        bgdot[q] = bg0 * dLp[q][0] + bg1 * dLp[q][1] + bg2 * dLp[q][2];
    }
    // tile-wide number of list entries that can still receive gradient
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxc = max(maxc, (uint32_t)__shfl_xor((int)maxc, d, 64));
    if (maxc == 0) {
        if (lane == 0) cutoff[tile] = make_uint2(0u, 0u);
        return;
    }
    if (lane == 0) {
        const uint32_t id = point_list[rg.x + maxc - 1];
        cutoff[tile] = make_uint2(__float_as_uint(xydr[id].z), id);
    }
    // gradient of pixel coordinate w.r.t. NDC (backward.cu:460-461)
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // walk the processed prefix [0, maxc) back to front, 64 instances at a time
    for (int hi = (int)maxc - 1; hi >= 0; hi -= 64) {
        const int cnt = min(64, hi + 1);
        uint32_t my_slot = 0;
        __syncthreads();
        if (lane < cnt) {
            const uint32_t id = point_list[rg.x + hi - lane];
            const float4 a = xydr[id];
            s_xy[lane] = make_float2(a.x, a.y);
            s_co[lane] = conic_opacity[id];
            s_rgb[lane] = rgb_clamped[id];
            // Gaussian-major slot of this (Gaussian, tile) instance: position in the
            // reference's duplicateWithKeys emission order (rasterizer_impl.cu:98-108)
            int x0, y0, x1, y1;
            tile_rect(a.x, a.y, (int)a.w, gx, gy, x0, y0, x1, y1);
            const uint32_t off = id == 0 ? 0u : point_offsets[id - 1];
            my_slot = off + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
        }
        __syncthreads();
        for (int k = 0; k < cnt; k++) {
            const int pos = hi - k;  // 0-based position in the tile list
            const float2 xy = s_xy[k];
            const float4 co = s_co[k];
            const float4 col = s_rgb[k];
            float part[FRG_SLOT_FLOATS];
#pragma unroll
            for (int c = 0; c < FRG_SLOT_FLOATS; c++) part[c] = 0.0f;
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if ((uint32_t)pos >= lastcon[q]) continue;
                float dx, dy;
                const float power = M::power(xy, co, pxf[q], pyf[q], dx, dy);
                if (power > 0.0f) continue;
                const float G = M::expo(power);
                const float alpha = fminf(0.99f, co.w * G);
                if (alpha < 1.0f / 255.0f) continue;
                any = true;
                Tr[q] = Tr[q] / (1.f - alpha);
                const float dchannel_dcolor = alpha * Tr[q];
                float dL_dalpha = 0.0f;
                const float cc[3] = {col.x, col.y, col.z};
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    accum[q][ch] = lasta[q] * lastc[q][ch] + (1.f - lasta[q]) * accum[q][ch];
                    lastc[q][ch] = cc[ch];
                    dL_dalpha += (cc[ch] - accum[q][ch]) * dLp[q][ch];
                    part[ch] += dchannel_dcolor * dLp[q][ch];
                }
                dL_dalpha *= Tr[q];
                lasta[q] = alpha;
                dL_dalpha += (-Tfin[q] / (1.f - alpha)) * bgdot[q];
                const float dL_dG = co.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co.x - gdy * co.y;
                const float dG_ddely = -gdy * co.z - gdx * co.y;
                part[3] += dL_dG * dG_ddelx * ddelx_dx;
                part[4] += dL_dG * dG_ddely * ddely_dy;
                part[5] += -0.5f * gdx * dx * dL_dG;
                part[6] += -0.5f * gdx * dy * dL_dG;
                part[7] += -0.5f * gdy * dy * dL_dG;
                part[8] += G * dL_dalpha;
            }
            if (__ballot(any) != 0ull) {
#pragma unroll
                for (int c = 0; c < FRG_SLOT_FLOATS; c++) part[c] = wave_sum_to_lane63(part[c]);
            }
            if (lane == 63) {
#pragma unroll
                for (int c = 0; c < FRG_SLOT_FLOATS; c++) s_part[k * FRG_SLOT_FLOATS + c] = part[c];
            }
        }
        __syncthreads();
        if (lane < cnt) {
            float* dst = slots + (size_t)my_slot * FRG_SLOT_FLOATS;
#pragma unroll
            for (int c = 0; c < FRG_SLOT_FLOATS; c++) dst[c] = s_part[lane * FRG_SLOT_FLOATS + c];
        }
    }
}

}  // namespace frg
