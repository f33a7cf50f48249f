// The Gaussian model's parameterisation, evaluated INSIDE the per-Gaussian kernels.
//
// The reference turns its optimizer's raw parameters into rasterizer inputs with a chain of eager torch
// kernels on every iteration:
//   opacity  = sigmoid(_opacities)                       frosting_model.py:726-727, gaussian_model.py:104-106
//   scale    = exp(_scales)                              frosting_model.py:32,763, gaussian_model.py:96-98
//   rotation = F.normalize(_quaternions)  (eps 1e-12)    frosting_model.py:797-798, gaussian_model.py:100-102
//   mean     = (bary_coords[..., None] * shell_cells_verts[_point_cell_indices].reshape(-1, 6, 3)).sum(-2),
//              bary_coords = softmax(_bary_coords) | relu(_bary_coords) / its sum (use_softmax_for_bary_coords)
//                                                        frosting_model.py:707-724 (Frosting's shell-bound centres)
// and autograd runs the chain backwards.  With RawInputs set, preprocess_fwd_kernel / preprocess_bwd_kernel read
// the raw parameters themselves: the activated tensors never exist in memory (2 x 44 bytes per Gaussian of
// traffic and four to ten launches per iteration less), and the backward emits gradients with respect to the
// raw parameters -- logits and, for a learnable shell (learn_shell = True), the cell vertices included.
#pragma once
#include <hip/hip_runtime.h>

namespace frg {

struct RawInputs {
    const float* raw_opacity = nullptr;       // [P]    (replaces opacities)
    const float* raw_scale = nullptr;         // [P,3]  (replaces scales)
    const float* raw_rot = nullptr;           // [P,4]  (replaces rotations)
    const float* shell_logits = nullptr;      // [P,6]  (replaces means3D, together with the two below)
    const float* shell_verts = nullptr;       // [F,6,3] = shell_cells_verts.reshape(-1, 6, 3): inner triangle, outer triangle
    const long long* shell_cells = nullptr;   // [P] _point_cell_indices
    int bary_mode = 0;                        // 0 softmax(logits) | 1 relu(x) / sum relu(x)  (use_softmax_for_bary_coords = False)
};

__device__ __forceinline__ float raw_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// softmax over six logits the way torch evaluates it (subtract the maximum)
__device__ __forceinline__ void raw_softmax6(const float* __restrict__ x, float* w)
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

// frosting_model.py:716-718 (use_softmax_for_bary_coords = False): relu, then divided by the sum -- no epsilon: the
// reference's expression, inf / nan for an all-negative row included
__device__ __forceinline__ void raw_relu_norm6(const float* __restrict__ x, float* w)
{
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) { w[k] = fmaxf(x[k], 0.0f); s += w[k]; }
#pragma unroll
    for (int k = 0; k < 6; k++) w[k] = w[k] / s;
}

__device__ __forceinline__ void raw_bary6(const RawInputs& r, int idx, float* w)
{
    if (r.bary_mode == 1) raw_relu_norm6(r.shell_logits + 6 * (size_t)idx, w);
    else raw_softmax6(r.shell_logits + 6 * (size_t)idx, w);
}

__device__ __forceinline__ float3 param_mean(const float* __restrict__ means3D, const RawInputs& r, int idx)
{
    if (!r.shell_logits) return make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float w[6];
    raw_bary6(r, idx, w);
    const float* v = r.shell_verts + 18 * (size_t)r.shell_cells[idx];
    float p[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) p[c] += w[k] * v[3 * k + c];
    return make_float3(p[0], p[1], p[2]);
}

__device__ __forceinline__ float3 param_scale(const float* __restrict__ scales, const RawInputs& r, int idx)
{
    if (!r.raw_scale) return make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
    return make_float3(expf(r.raw_scale[3 * idx]), expf(r.raw_scale[3 * idx + 1]), expf(r.raw_scale[3 * idx + 2]));
}

__device__ __forceinline__ float4 param_rot(const float* __restrict__ rotations, const RawInputs& r, int idx)
{
    if (!r.raw_rot) return *reinterpret_cast<const float4*>(rotations + 4 * idx);
    const float4 q = make_float4(r.raw_rot[4 * idx], r.raw_rot[4 * idx + 1], r.raw_rot[4 * idx + 2], r.raw_rot[4 * idx + 3]);
    const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    return make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
}

__device__ __forceinline__ float param_opacity(const float* __restrict__ opacities, const RawInputs& r, int idx)
{
    return r.raw_opacity ? raw_sigmoid(r.raw_opacity[idx]) : opacities[idx];
}

}  // namespace frg
