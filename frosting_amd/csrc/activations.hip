// Parameter activations of the Gaussian model, forward and backward, one launch each
// (part of SURVEY.md 8(f) rank 3: the eager elementwise chain between the optimizer's raw
// parameters and the rasterizer's inputs).
//
// Reference: gaussian_splatting/scene/gaussian_model.py:32-40,96-115 and
// frosting_scene/frosting_model.py:32,726,797-798 --
//   opacity  = sigmoid(raw)            scale = exp(raw)            rotation = F.normalize(raw)  (eps 1e-12)
// The backward takes the rasterizer's gradients w.r.t. the activated values and overwrites them, in
// place, with the gradients w.r.t. the raw parameters, so that the flat gradient buffer can go
// straight into frg_adam_step.
#include "kernels.h"

namespace frg {

__global__ void __launch_bounds__(256)
activate_kernel(int P, const float* __restrict__ raw_opacity, const float* __restrict__ raw_scale,
                const float* __restrict__ raw_rot, float* __restrict__ opacity, float* __restrict__ scale,
                float* __restrict__ rot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    opacity[i] = 1.0f / (1.0f + expf(-raw_opacity[i]));
#pragma unroll
    for (int c = 0; c < 3; c++) scale[3 * i + c] = expf(raw_scale[3 * i + c]);
    // (scalar accesses: inside a flat parameter buffer the rotation block starts at an arbitrary offset)
    const float4 q = make_float4(raw_rot[4 * i], raw_rot[4 * i + 1], raw_rot[4 * i + 2], raw_rot[4 * i + 3]);
    const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    rot[4 * i] = q.x / n; rot[4 * i + 1] = q.y / n; rot[4 * i + 2] = q.z / n; rot[4 * i + 3] = q.w / n;
}

// g_* enter as dL/d(activated) and leave as dL/d(raw)
__global__ void __launch_bounds__(256)
activate_bwd_kernel(int P, const float* __restrict__ opacity, const float* __restrict__ scale,
                    const float* __restrict__ raw_rot, float* __restrict__ g_opacity, float* __restrict__ g_scale,
                    float* __restrict__ g_rot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float o = opacity[i];
    g_opacity[i] = g_opacity[i] * ((1.0f - o) * o);                       // sigmoid'
#pragma unroll
    for (int c = 0; c < 3; c++) g_scale[3 * i + c] = g_scale[3 * i + c] * scale[3 * i + c];   // exp'
    // y = x / max(|x|, eps):  dx = (g - y (y . g)) / max(|x|, eps)   (|x| > eps)
    const float4 q = make_float4(raw_rot[4 * i], raw_rot[4 * i + 1], raw_rot[4 * i + 2], raw_rot[4 * i + 3]);
    const float4 g = make_float4(g_rot[4 * i], g_rot[4 * i + 1], g_rot[4 * i + 2], g_rot[4 * i + 3]);
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float n = fmaxf(nrm, 1e-12f);
    const float inv = 1.0f / n;
    const float4 y = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    // below eps the denominator is the constant eps: plain scaling, no projection
    const float d = nrm > 1e-12f ? (y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w) : 0.0f;
    g_rot[4 * i] = (g.x - y.x * d) * inv; g_rot[4 * i + 1] = (g.y - y.y * d) * inv;
    g_rot[4 * i + 2] = (g.z - y.z * d) * inv; g_rot[4 * i + 3] = (g.w - y.w * d) * inv;
}

hipError_t launch_activate(int P, const float* raw_opacity, const float* raw_scale, const float* raw_rot, float* opacity,
                           float* scale, float* rot, hipStream_t s)
{
    hipLaunchKernelGGL(activate_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, raw_opacity, raw_scale, raw_rot, opacity,
                       scale, rot);
    return hipGetLastError();
}

hipError_t launch_activate_bwd(int P, const float* opacity, const float* scale, const float* raw_rot, float* g_opacity,
                               float* g_scale, float* g_rot, hipStream_t s)
{
    hipLaunchKernelGGL(activate_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, opacity, scale, raw_rot, g_opacity,
                       g_scale, g_rot);
    return hipGetLastError();
}

}  // namespace frg
