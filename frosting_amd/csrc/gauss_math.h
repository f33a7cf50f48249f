// Per-Gaussian math shared by the forward and backward per-Gaussian kernels.
// Evaluated as IEEE binary32 with NO contraction, in the association order of
// the reference (forward.cu:20-152, auxiliary.h:41-77) -- this is what makes
// radii, tile rectangles and depth keys bit-identical.  Translation units that
// include this header are compiled with -ffp-contract=off.
#pragma once
#include "frg_common.h"

#pragma clang fp contract(off)

namespace frg {

__device__ constexpr float kSH0 = 0.28209479177387814f;
__device__ constexpr float kSH1 = 0.4886025119029199f;
__device__ constexpr float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                      -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                      0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                      -0.5900435899266435f};

// p * M for the reference's row-vector matrices (column-major read)
__device__ __forceinline__ float3 xform43(const float3 p, const float* m)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform44(const float3 p, const float* m)
{
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
                       m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

// ((v + 1) * S - 1) / 2 in double, one rounding to float (auxiliary.h:41-44)
__device__ __forceinline__ float ndc_to_pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

// Rotation (columns R[c], glm layout) from the quaternion AS GIVEN (r,x,y,z);
// the reference does not normalise (forward.cu:127).
struct Rot3 { float c[3][3]; };
__device__ __forceinline__ Rot3 quat_to_rot(const float4 q)
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    Rot3 R;
    R.c[0][0] = 1.f - 2.f * (y * y + z * z); R.c[0][1] = 2.f * (x * y - r * z); R.c[0][2] = 2.f * (x * z + r * y);
    R.c[1][0] = 2.f * (x * y + r * z); R.c[1][1] = 1.f - 2.f * (x * x + z * z); R.c[1][2] = 2.f * (y * z - r * x);
    R.c[2][0] = 2.f * (x * z - r * y); R.c[2][1] = 2.f * (y * z + r * x); R.c[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

// Sigma = (S R)^T (S R), upper triangle (forward.cu:118-152)
__device__ __forceinline__ void cov3d_from_scale_rot(const float3 scale, float mod, const float4 q, float* cov)
{
    const float s[3] = {mod * scale.x, mod * scale.y, mod * scale.z};
    const Rot3 R = quat_to_rot(q);
    float M[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) M[c][r] = s[r] * R.c[c][r];
#define FRG_SIG(c, r) (M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2])
    cov[0] = FRG_SIG(0, 0); cov[1] = FRG_SIG(0, 1); cov[2] = FRG_SIG(0, 2);
    cov[3] = FRG_SIG(1, 1); cov[4] = FRG_SIG(1, 2); cov[5] = FRG_SIG(2, 2);
#undef FRG_SIG
}

// EWA projection set-up (forward.cu:74-100, backward.cu:165-195): clamped view-space
// point t and T = W J with T[c][r] (third column is zero).
struct Ewa { float t[3]; float T[2][3]; float xmul, ymul; };
__device__ __forceinline__ Ewa ewa_setup(const float3 mean, float fx, float fy, float tan_fovx, float tan_fovy, const float* vm)
{
    Ewa e;
    float3 t = xform43(mean, vm);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
    const float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float W0 = vm[4 * r + 0], W1 = vm[4 * r + 1], W2 = vm[4 * r + 2];
        e.T[0][r] = W0 * J00 + W1 * 0.0f + W2 * J02;
        e.T[1][r] = W0 * 0.0f + W1 * J11 + W2 * J12;
    }
    e.t[0] = t.x; e.t[1] = t.y; e.t[2] = t.z;
    return e;
}
// cov2D = T^t Vrk^t T: entries (0,0),(0,1),(1,1) before the +0.3 low-pass
__device__ __forceinline__ void ewa_cov2d(const Ewa& e, const float* c3, float& a, float& b, float& c)
{
    const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float X[3][2];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int r = 0; r < 2; r++) X[k][r] = e.T[r][0] * V[0][k] + e.T[r][1] * V[1][k] + e.T[r][2] * V[2][k];
    a = X[0][0] * e.T[0][0] + X[1][0] * e.T[0][1] + X[2][0] * e.T[0][2];
    b = X[0][1] * e.T[0][0] + X[1][1] * e.T[0][1] + X[2][1] * e.T[0][2];
    c = X[0][1] * e.T[1][0] + X[1][1] * e.T[1][1] + X[2][1] * e.T[1][2];
}

// SH basis weights w[i] such that colour = sum_i w[i] * sh[i] in the reference's
// left-to-right order (forward.cu:20-71).  Returns the number of active terms.
__device__ __forceinline__ int sh_weights(int deg, float x, float y, float z, float* w)
{
    w[0] = kSH0;
    if (deg < 1) return 1;
    w[1] = kSH1 * y; w[2] = kSH1 * z; w[3] = kSH1 * x;   // applied as -,+,-
    if (deg < 2) return 4;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    w[4] = kSH2[0] * xy; w[5] = kSH2[1] * yz; w[6] = kSH2[2] * (2.0f * zz - xx - yy);
    w[7] = kSH2[3] * xz; w[8] = kSH2[4] * (xx - yy);
    if (deg < 3) return 9;
    w[9] = kSH3[0] * y * (3.0f * xx - yy); w[10] = kSH3[1] * xy * z;
    w[11] = kSH3[2] * y * (4.0f * zz - xx - yy);
    w[12] = kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    w[13] = kSH3[4] * x * (4.0f * zz - xx - yy); w[14] = kSH3[5] * z * (xx - yy);
    w[15] = kSH3[6] * x * (xx - 3.0f * yy);
    return 16;
}

// d(colour_ch)/d(dir): the share of one coefficient (basis i, value sv of the channel) in
//   ddx = sum_i dbasis_i/dx * sh[i][ch], ddy, ddz                       (backward.cu:47-137).
// Evaluated by the FORWARD while the coefficients pass through LDS anyway, and stored (9 floats per Gaussian,
// GeomState::sh_dir): the backward then never reads the 192-byte SH rows again.  i is a compile-time constant at
// every call site (unrolled), so only one case survives.
struct ShDir {
    float x, y, z, xx, yy, zz, xy, yz, xz;
    int deg;
    __device__ __forceinline__ ShDir(int deg_, float x_, float y_, float z_)
        : x(x_), y(y_), z(z_), xx(x_ * x_), yy(y_ * y_), zz(z_ * z_), xy(x_ * y_), yz(y_ * z_), xz(x_ * z_), deg(deg_) {}
    __device__ __forceinline__ void feed(int i, float sv, float& ddx, float& ddy, float& ddz) const
    {
        if (i >= 1 && i <= 3 && deg < 1) return;
        if (i >= 4 && i <= 8 && deg < 2) return;
        if (i >= 9 && deg < 3) return;
        switch (i) {
        case 1: ddy += -kSH1 * sv; break;
        case 2: ddz += kSH1 * sv; break;
        case 3: ddx += -kSH1 * sv; break;
        case 4: ddx += kSH2[0] * y * sv; ddy += kSH2[0] * x * sv; break;
        case 5: ddy += kSH2[1] * z * sv; ddz += kSH2[1] * y * sv; break;
        case 6: ddx += kSH2[2] * 2.f * -x * sv; ddy += kSH2[2] * 2.f * -y * sv; ddz += kSH2[2] * 2.f * 2.f * z * sv; break;
        case 7: ddx += kSH2[3] * z * sv; ddz += kSH2[3] * x * sv; break;
        case 8: ddx += kSH2[4] * 2.f * x * sv; ddy += kSH2[4] * 2.f * -y * sv; break;
        case 9: ddx += kSH3[0] * sv * 3.f * 2.f * xy; ddy += kSH3[0] * sv * 3.f * (xx - yy); break;
        case 10: ddx += kSH3[1] * sv * yz; ddy += kSH3[1] * sv * xz; ddz += kSH3[1] * sv * xy; break;
        case 11: ddx += kSH3[2] * sv * -2.f * xy; ddy += kSH3[2] * sv * (-3.f * yy + 4.f * zz - xx); ddz += kSH3[2] * sv * 4.f * 2.f * yz; break;
        case 12: ddx += kSH3[3] * sv * -3.f * 2.f * xz; ddy += kSH3[3] * sv * -3.f * 2.f * yz; ddz += kSH3[3] * sv * 3.f * (2.f * zz - xx - yy); break;
        case 13: ddx += kSH3[4] * sv * (-3.f * xx + 4.f * zz - yy); ddy += kSH3[4] * sv * -2.f * xy; ddz += kSH3[4] * sv * 4.f * 2.f * xz; break;
        case 14: ddx += kSH3[5] * sv * 2.f * xz; ddy += kSH3[5] * sv * -2.f * yz; ddz += kSH3[5] * sv * (xx - yy); break;
        case 15: ddx += kSH3[6] * sv * 3.f * (xx - yy); ddy += kSH3[6] * sv * -3.f * 2.f * xy; break;
        default: break;
        }
    }
};

}  // namespace frg
