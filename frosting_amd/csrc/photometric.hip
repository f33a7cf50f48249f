// Fused photometric loss, forward and backward (the step right before the
// rasterizer's backward).
//
// Replaces   (1 - lambda) * l1_loss(pred, gt) + lambda * (1 - ssim(pred, gt))
// of frosting_utils/loss_utils.py:17-62 as the refinement uses it (frosting_trainers/refine.py:407-409,
// lambda = 0.2): there, six grouped 11x11 conv2d calls over [3,H,W] plus a dozen elementwise kernels
// and their autograd temporaries.  Here two tiled kernels: the first forms the five windowed moments
// (mu1, mu2, E[x^2], E[y^2], E[xy]) with a separable pass through LDS, evaluates the SSIM map and keeps
// its three partial derivatives per pixel; the second convolves those three maps with the same window
// and assembles dL/dpred, L1 term included.  The scalar loss is reduced in two fixed-order stages
// (bit-reproducible).  Zero padding like conv2d(padding = 5).
#include "kernels.h"

namespace frg {

#define PH_TILE 16
#define PH_R 5                       // window radius (window_size 11)
#define PH_HALO (PH_TILE + 2 * PH_R) // 26

struct PhotoWindow { float w[2 * PH_R + 1]; };

__device__ __forceinline__ float block_sum_256(float v, float* red)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

// grid (ceil(W/16), ceil(H/16), C), 256 threads
__global__ void __launch_bounds__(256)
photometric_fwd_kernel(int W, int H, const float* __restrict__ pred, const float* __restrict__ gt, PhotoWindow win,
                       float* __restrict__ m_mu, float* __restrict__ m_xx, float* __restrict__ m_xy,
                       float* __restrict__ partial)
{
    __shared__ float sx[PH_HALO][PH_HALO + 1], sy[PH_HALO][PH_HALO + 1];
    __shared__ float hz[5][PH_HALO][PH_TILE + 1];
    __shared__ float red[4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * PH_TILE, y0 = blockIdx.y * PH_TILE;
    const size_t plane = (size_t)W * H, off = (size_t)blockIdx.z * plane;
    for (int i = threadIdx.x; i < PH_HALO * PH_HALO; i += 256) {
        const int r = i / PH_HALO, c = i - r * PH_HALO;
        const int gx = x0 + c - PH_R, gy = y0 + r - PH_R;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        sx[r][c] = in ? pred[off + (size_t)gy * W + gx] : 0.0f;
        sy[r][c] = in ? gt[off + (size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    // horizontal pass: 26 rows x 16 columns, five moments
    for (int i = threadIdx.x; i < PH_HALO * PH_TILE; i += 256) {
        const int r = i / PH_TILE, c = i - r * PH_TILE;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * PH_R + 1; k++) {
            const float x = sx[r][c + k], y = sy[r][c + k], w = win.w[k];
            a0 += w * x; a1 += w * y; a2 += w * (x * x); a3 += w * (y * y); a4 += w * (x * y);
        }
        hz[0][r][c] = a0; hz[1][r][c] = a1; hz[2][r][c] = a2; hz[3][r][c] = a3; hz[4][r][c] = a4;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * PH_R + 1; k++) {
        const float w = win.w[k];
        mu1 += w * hz[0][ty + k][tx]; mu2 += w * hz[1][ty + k][tx];
        exx += w * hz[2][ty + k][tx]; eyy += w * hz[3][ty + k][tx]; exy += w * hz[4][ty + k][tx];
    }
    const int px = x0 + tx, py = y0 + ty;
    const bool inside = px < W && py < H;
    float l1 = 0.f, ssim = 0.f;
    if (inside) {
        // loss_utils.py:49-58
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = exx - mu1_sq, s2 = eyy - mu2_sq, s12 = exy - mu12;
        const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
        const float inv = 1.0f / (Cc * D);
        ssim = A * B * inv;
        // partial derivatives of the map w.r.t. the windowed moments of pred (mu1, E[x^2], E[xy])
        const float dA = B * inv, dB = A * inv, dC = -ssim / Cc, dD = -ssim / D;
        const size_t p = off + (size_t)py * W + px;
        m_mu[p] = dA * 2.f * mu2 + dC * 2.f * mu1 - dB * 2.f * mu2 - dD * 2.f * mu1;
        m_xx[p] = dD;
        m_xy[p] = 2.f * dB;
        l1 = fabsf(sx[ty + PH_R][tx + PH_R] - sy[ty + PH_R][tx + PH_R]);
    }
    const float bl1 = block_sum_256(l1, red);
    const float bss = block_sum_256(ssim, red);
    if (threadIdx.x == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * b] = bl1;
        partial[2 * b + 1] = bss;
    }
}

// one workgroup: fixed-order sum of the per-tile partials -> loss
__global__ void __launch_bounds__(256)
photometric_finalize_kernel(int nblocks, const float* __restrict__ partial, float inv_count, float lambda, float* __restrict__ loss)
{
    __shared__ float red[4];
    float l1 = 0.f, ss = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) { l1 += partial[2 * b]; ss += partial[2 * b + 1]; }
    const float tl1 = block_sum_256(l1, red);
    const float tss = block_sum_256(ss, red);
    if (threadIdx.x == 0) loss[0] = (1.0f - lambda) * (tl1 * inv_count) + lambda * (1.0f - tss * inv_count);
}

// dL/dpred = (1 - lambda)/count * sign(pred - gt) - lambda/count * (W * m_mu + 2 pred (W * m_xx) + gt (W * m_xy))
__global__ void __launch_bounds__(256)
photometric_bwd_kernel(int W, int H, const float* __restrict__ pred, const float* __restrict__ gt, PhotoWindow win,
                       const float* __restrict__ m_mu, const float* __restrict__ m_xx, const float* __restrict__ m_xy,
                       float inv_count, float lambda, float* __restrict__ dL_dpred)
{
    __shared__ float sm[3][PH_HALO][PH_HALO + 1];
    __shared__ float hz[3][PH_HALO][PH_TILE + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * PH_TILE, y0 = blockIdx.y * PH_TILE;
    const size_t plane = (size_t)W * H, off = (size_t)blockIdx.z * plane;
    for (int i = threadIdx.x; i < PH_HALO * PH_HALO; i += 256) {
        const int r = i / PH_HALO, c = i - r * PH_HALO;
        const int gx = x0 + c - PH_R, gy = y0 + r - PH_R;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const size_t p = off + (size_t)gy * W + gx;
        sm[0][r][c] = in ? m_mu[p] : 0.0f;
        sm[1][r][c] = in ? m_xx[p] : 0.0f;
        sm[2][r][c] = in ? m_xy[p] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PH_HALO * PH_TILE; i += 256) {
        const int r = i / PH_TILE, c = i - r * PH_TILE;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * PH_R + 1; k++) {
            const float w = win.w[k];
            a0 += w * sm[0][r][c + k]; a1 += w * sm[1][r][c + k]; a2 += w * sm[2][r][c + k];
        }
        hz[0][r][c] = a0; hz[1][r][c] = a1; hz[2][r][c] = a2;
    }
    __syncthreads();
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * PH_R + 1; k++) {
        const float w = win.w[k];
        g0 += w * hz[0][ty + k][tx]; g1 += w * hz[1][ty + k][tx]; g2 += w * hz[2][ty + k][tx];
    }
    const int px = x0 + tx, py = y0 + ty;
    if (px < W && py < H) {
        const size_t p = off + (size_t)py * W + px;
        const float x = pred[p], y = gt[p];
        const float d = x - y;
        const float sgn = d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f);
        dL_dpred[p] = (1.0f - lambda) * inv_count * sgn - lambda * inv_count * (g0 + 2.f * x * g1 + y * g2);
    }
}

size_t photometric_workspace_bytes(int C, int W, int H)
{
    const size_t n = (size_t)C * W * H;
    const size_t blocks = (size_t)C * ((W + PH_TILE - 1) / PH_TILE) * ((H + PH_TILE - 1) / PH_TILE);
    return align_up(3 * n * 4, 256) + align_up(2 * blocks * 4, 256);
}

hipError_t launch_photometric(int C, int W, int H, const float* pred, const float* gt, const float* window11, float lambda,
                              float* loss, float* dL_dpred, char* workspace, hipStream_t s)
{
    const size_t n = (size_t)C * W * H;
    float* m_mu = reinterpret_cast<float*>(workspace);
    float* m_xx = m_mu + n;
    float* m_xy = m_xx + n;
    float* partial = reinterpret_cast<float*>(workspace + align_up(3 * n * 4, 256));
    PhotoWindow win;
    for (int k = 0; k < 2 * PH_R + 1; k++) win.w[k] = window11[k];
    const dim3 grid((W + PH_TILE - 1) / PH_TILE, (H + PH_TILE - 1) / PH_TILE, C), block(256);
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    const float inv_count = 1.0f / (float)n;
    hipLaunchKernelGGL(photometric_fwd_kernel, grid, block, 0, s, W, H, pred, gt, win, m_mu, m_xx, m_xy, partial);
    hipLaunchKernelGGL(photometric_finalize_kernel, dim3(1), dim3(256), 0, s, nblocks, partial, inv_count, lambda, loss);
    if (dL_dpred)
        hipLaunchKernelGGL(photometric_bwd_kernel, grid, block, 0, s, W, H, pred, gt, win, m_mu, m_xx, m_xy, inv_count, lambda,
                           dL_dpred);
    return hipGetLastError();
}

}  // namespace frg
