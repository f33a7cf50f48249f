// Frosting's shell parameterisation of the Gaussian centres, forward and backward (the rest of
// SURVEY.md 8(f) rank 3).
//
// Reference: frosting_scene/frosting_model.py:707-724 --
//     bary   = softmax(_bary_coords, dim=-1)                                   [P,6]
//     points = (bary[..., None] * shell_cells_verts[_point_cell_indices].reshape(-1, 6, 3)).sum(-2)
// with shell_cells_verts [F,2,3,3] = the inner and the outer triangle of every prismatic cell.  In the
// default configuration (learn_shell = False, frosting_model.py:2325) the cell vertices are constants
// and only the six logits per Gaussian are trained: one thread per Gaussian gathers the cell's 72 bytes,
// forms the softmax in torch's way (subtract the maximum) and the weighted sum; the backward recomputes
// the weights and applies the softmax Jacobian to the six dot products v_k . dL/dpoint.
#include "kernels.h"

namespace frg {

__device__ __forceinline__ void softmax6(const float* x, float* w)
{
    float m = x[0];
#pragma unroll
    for (int k = 1; k < 6; k++) m = fmaxf(m, x[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) { w[k] = expf(x[k] - m); s += w[k]; }
#pragma unroll
    for (int k = 0; k < 6; k++) w[k] = w[k] / s;
}

__global__ void __launch_bounds__(256)
shell_points_kernel(int P, const float* __restrict__ logits, const float* __restrict__ cell_verts,
                    const long long* __restrict__ cell, float* __restrict__ points)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float x[6], w[6];
#pragma unroll
    for (int k = 0; k < 6; k++) x[k] = logits[6 * (size_t)i + k];
    softmax6(x, w);
    const float* v = cell_verts + 18 * (size_t)cell[i];
    float p[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) p[c] += w[k] * v[3 * k + c];
    points[3 * (size_t)i] = p[0]; points[3 * (size_t)i + 1] = p[1]; points[3 * (size_t)i + 2] = p[2];
}

__global__ void __launch_bounds__(256)
shell_points_bwd_kernel(int P, const float* __restrict__ logits, const float* __restrict__ cell_verts,
                        const long long* __restrict__ cell, const float* __restrict__ dL_dpoints,
                        float* __restrict__ dL_dlogits)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float x[6], w[6], g[6];
#pragma unroll
    for (int k = 0; k < 6; k++) x[k] = logits[6 * (size_t)i + k];
    softmax6(x, w);
    const float* v = cell_verts + 18 * (size_t)cell[i];
    const float d0 = dL_dpoints[3 * (size_t)i], d1 = dL_dpoints[3 * (size_t)i + 1], d2 = dL_dpoints[3 * (size_t)i + 2];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        g[k] = v[3 * k] * d0 + v[3 * k + 1] * d1 + v[3 * k + 2] * d2;    // dL/dw_k
        dot += w[k] * g[k];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dlogits[6 * (size_t)i + k] = w[k] * (g[k] - dot);   // softmax Jacobian
}

hipError_t launch_shell_points(int P, const float* logits, const float* cell_verts, const long long* cell, float* points,
                               hipStream_t s)
{
    hipLaunchKernelGGL(shell_points_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, logits, cell_verts, cell, points);
    return hipGetLastError();
}

hipError_t launch_shell_points_bwd(int P, const float* logits, const float* cell_verts, const long long* cell,
                                   const float* dL_dpoints, float* dL_dlogits, hipStream_t s)
{
    hipLaunchKernelGGL(shell_points_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, logits, cell_verts, cell, dL_dpoints,
                       dL_dlogits);
    return hipGetLastError();
}

}  // namespace frg
