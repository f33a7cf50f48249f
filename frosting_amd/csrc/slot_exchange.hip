// Slot-sum exchange of the view-parallel step (no counterpart in the single-GPU reference).
//
// After phase 1 of the backward (blend backward + slot reduction) everything one view contributes to the gradient of a
// Gaussian is determined by NINE floats -- the clamp-masked colour gradient and the six pixel moments of G dL/dalpha
// (preprocess_bwd.hip, "sums") -- together with data every rank already holds: the parameters and the view's camera.
// And only the Gaussians some pixel reached before its tile saturated have sums at all (one in eight at C3).  So the
// ranks exchange those sums instead of finished gradients:
//
//   pack     the rows {dRGB[3], moments[6]} (36 bytes) of the Gaussians with a gradient, IN INDEX ORDER, behind a bit mask
//            (one bit per Gaussian) and one row offset per block of 64 Gaussians: Gaussian g finds its row in view v as
//            base[v][g / 64] + popcount(mask[v][g / 64] below g).  Fixed capacity: no host wait for a count.
//   gather   one all-gather of the packets (host side: frosting_amd/parallel.py).
//   combine  ONE pass over the Gaussians: for every view that has a row, in VIEW ORDER, the per-Gaussian backward chain of
//            preprocess_bwd.hip (sections 2-5: cov2D, projection, SH, cov3D -> scale / quaternion) with that view's camera,
//            the 59 gradient floats accumulated in registers and every row written once -- the single-process accumulation
//            of the per-view gradients, bit for bit (same expressions, no contraction, same order of additions), without
//            the dense zero fills, per-view scatters and SH rebuild of the round-5 plans.
//
// What a view's geometry record held for phase 2 -- conic and opacity -- is recomputed here from the parameters with the
// forward's own functions (gauss_math.h), bit-identically; d(colour)/d(direction) is formed from the SH row as the
// sh_dir_in_backward form of preprocess_bwd.hip does (the same bits as the forward's: tests/test_gpu_parity.py).
#include "gauss_math.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace frg {

// ---- packet layout (uint32 words) -------------------------------------------------------------------------------------------
//   [0] rows packed  [1] rows wanted (> capacity: overflow)  [2] Gaussians of the packet  [3] capacity  [4] first Gaussian
//   [5] magic  [8..23] viewmatrix  [24..39] projmatrix  [40..42] camera centre  [43] tan_fovx  [44] tan_fovy
//   [45] width  [46] height  [47] scale_modifier  [48] active SH degree
//   masks   uint64[nblk]  at word 64                 (nblk = blocks of 64 Gaussians)
//   bases   uint32[nblk]  behind them, 16-byte aligned
//   rows    float[capacity][9] behind them, 16-byte aligned
__host__ __device__ inline size_t sum_packet_blocks(size_t n) { return (n + 63) / 64; }
__host__ __device__ inline size_t sum_packet_bases_word(size_t n) { return FRG_SUM_HDR_WORDS + 2 * sum_packet_blocks(n); }
__host__ __device__ inline size_t sum_packet_rows_word(size_t n) { return (sum_packet_bases_word(n) + sum_packet_blocks(n) + 3) / 4 * 4; }
size_t sum_packet_bytes(size_t n, size_t capacity) { return ((sum_packet_rows_word(n) + FRG_SUM_ROW_FLOATS * capacity + 3) / 4 * 4) * 4; }

// One workgroup: the packet's masks (copied from the phase-1 workspace), the exclusive prefix of their popcounts, the header.
__global__ void __launch_bounds__(1024)
sum_rows_scan_kernel(int first, int n, uint32_t capacity, const unsigned long long* __restrict__ live_masks,
                     uint32_t* __restrict__ packet, SumCamera cam, const float* __restrict__ viewmatrix,
                     const float* __restrict__ projmatrix, const float* __restrict__ campos)
{
    __shared__ uint32_t part[1024];
    const int nblk = (int)sum_packet_blocks((size_t)n), tid = threadIdx.x;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(packet + FRG_SUM_HDR_WORDS);
    uint32_t* bases = packet + sum_packet_bases_word((size_t)n);
    const int per = (nblk + 1023) / 1024, b0 = tid * per, b1 = min(nblk, b0 + per);
    const unsigned long long* src = live_masks + first / 64;
    uint32_t sum = 0;
    for (int b = b0; b < b1; b++) sum += (uint32_t)__popcll(src[b]);
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {           // inclusive scan of the threads' totals
        const uint32_t up = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += up;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (int b = b0; b < b1; b++) {
        const unsigned long long m = src[b];
        masks[b] = m;
        bases[b] = run;
        run += (uint32_t)__popcll(m);
    }
    if (tid == 0) {
        const uint32_t want = part[1023];
        packet[0] = want < capacity ? want : capacity;
        packet[1] = want;
        packet[2] = (uint32_t)n; packet[3] = capacity; packet[4] = (uint32_t)first; packet[5] = FRG_SUM_MAGIC;
        packet[6] = 0u; packet[7] = 0u;
        float* f = reinterpret_cast<float*>(packet);
        for (int i = 0; i < 16; i++) { f[8 + i] = viewmatrix[i]; f[24 + i] = projmatrix[i]; }
        f[40] = campos[0]; f[41] = campos[1]; f[42] = campos[2];
        f[43] = cam.tan_fovx; f[44] = cam.tan_fovy;
        packet[45] = (uint32_t)cam.width; packet[46] = (uint32_t)cam.height;
        f[47] = cam.scale_modifier; packet[48] = (uint32_t)cam.D;
        for (int i = 49; i < FRG_SUM_HDR_WORDS; i++) packet[i] = 0u;
    }
}

// One wave per block of 64 Gaussians: the rows of the marked ones, in index order.
__global__ void __launch_bounds__(256)
sum_rows_pack_kernel(int first, int n, uint32_t capacity, const float* __restrict__ sums, const float* __restrict__ drgb_masked,
                     uint32_t* __restrict__ packet)
{
    const int lane = threadIdx.x & 63, blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= (int)sum_packet_blocks((size_t)n)) return;
    const unsigned long long m = reinterpret_cast<const unsigned long long*>(packet + FRG_SUM_HDR_WORDS)[blk];
    if (!((m >> lane) & 1ull)) return;
    const uint32_t row = (packet + sum_packet_bases_word((size_t)n))[blk] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (row >= capacity) return;                 // (over capacity: the header says so; the host packs again into a larger packet)
    const size_t g = (size_t)first + (size_t)blk * 64 + lane;
    float* dst = reinterpret_cast<float*>(packet) + sum_packet_rows_word((size_t)n) + (size_t)row * FRG_SUM_ROW_FLOATS;
    const float* s = sums + g * FRG_SLOT_FLOATS;
    dst[0] = drgb_masked[3 * g]; dst[1] = drgb_masked[3 * g + 1]; dst[2] = drgb_masked[3 * g + 2];
#pragma unroll
    for (int c = 3; c < FRG_SLOT_FLOATS; c++) dst[c] = s[c];
}

hipError_t launch_pack_sum_rows(int first, int n, uint32_t capacity, const unsigned long long* live_masks, const float* sums,
                                const float* drgb_masked, const SumCamera& cam, const float* viewmatrix, const float* projmatrix,
                                const float* campos, void* packet, hipStream_t s)
{
    uint32_t* pk = reinterpret_cast<uint32_t*>(packet);
    hipLaunchKernelGGL(sum_rows_scan_kernel, dim3(1), dim3(1024), 0, s, first, n, capacity, live_masks, pk, cam, viewmatrix, projmatrix, campos);
    const int nblk = (int)sum_packet_blocks((size_t)n);
    hipLaunchKernelGGL(sum_rows_pack_kernel, dim3((nblk + 3) / 4), dim3(256), 0, s, first, n, capacity, sums, drgb_masked, pk);
    return hipGetLastError();
}

// ---- combine ------------------------------------------------------------------------------------------------------------------
#define CMB_THREADS 256
#define CMB_TILE 1024                   // Gaussians per workgroup: the ones with a row in some view are compacted over the tile
#define CMB_MAX_VIEWS 16

struct CmbCam { float view[16], proj[16], campos[3], tan_fovx, tan_fovy, focal_x, focal_y, half_w, half_h, scale_modifier; int D; };

// sections 2 - 5 of preprocess_bwd_kernel for ONE (Gaussian, view): `part` = the view's nine slot sums of the Gaussian, the
// colour part already clamp-masked.  Adds the view's gradient to the accumulators.  Expression for expression the chain of
// preprocess_bwd.hip (has_grad branch, SH16, the backward forms d(colour)/d(direction)); tests pin the two bit for bit.
// raw-parameter mode (raw_params.h): the activations' Jacobians are applied per view, as phase 2 applies them.
__device__ __forceinline__ void combine_one_view(const CmbCam& cm, const float3 mean, const float3 sc, const float4 q, const float o,
                                                 const bool raw_opacity, const bool raw_scale, const bool raw_rot, const float4 q_raw,
                                                 const float4* __restrict__ sh_row, float (&part)[FRG_SLOT_FLOATS],
                                                 float (&a_mean)[3], float (&a_scale)[3], float (&a_rot)[4], float& a_opac, float (&a_sh)[48])
{
    // the forward's conic (preprocess.hip preprocess_one): cov3D -> EWA cov2D -> + 0.3 -> inverse
    float cov[6];
    cov3d_from_scale_rot(sc, cm.scale_modifier, q, cov);
    const Ewa e = ewa_setup(mean, cm.focal_x, cm.focal_y, cm.tan_fovx, cm.tan_fovy, cm.view);
    float a, b, c;
    ewa_cov2d(e, cov, a, b, c);
    a += 0.3f; c += 0.3f;
    const float denom = a * c - b * b;
    const float det_inv = 1.f / denom;
    const float4 kc = make_float4(c * det_inv, -b * det_inv, a * det_inv, o);
    // d(colour)/d(direction) from the SH row, coefficient after coefficient as the forward's SH pass adds them
    const float dox = mean.x - cm.campos[0], doy = mean.y - cm.campos[1], doz = mean.z - cm.campos[2];
    const float len = sqrtf(dox * dox + doy * doy + doz * doz);
    const float x = dox / len, y = doy / len, z = doz / len;
    float shd[9];
#pragma unroll
    for (int k = 0; k < 9; k++) shd[k] = 0.0f;
    {
        const ShDir sd(cm.D, x, y, z);
        const int ncoef = (cm.D + 1) * (cm.D + 1);
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const float4 v = sh_row[j];
            const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int ee = 4 * j + t, i = ee / 3, ch = ee % 3;
                if (i < ncoef) sd.feed(i, f[t], shd[ch], shd[3 + ch], shd[6 + ch]);
            }
        }
    }
    // pixel moments -> the reference's terms (backward.cu:536-554), once per Gaussian
    {
        const float m3 = part[3], m4 = part[4];
        part[3] = -o * (kc.x * m3 + kc.y * m4) * cm.half_w;
        part[4] = -o * (kc.z * m4 + kc.y * m3) * cm.half_h;
        part[5] = -0.5f * o * part[5];
        part[6] = -0.5f * o * part[6];
        part[7] = -0.5f * o * part[7];
    }
    float dmean[3], dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- computeCov2DCUDA (backward.cu:144-274) ----
    {
        const float dLc0 = part[5], dLc1 = part[6], dLc3 = part[7];
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define T_(cc, rr) e.T[cc][rr]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dLc0 + 2 * b * c * dLc1 + (denom - a * c) * dLc3);
            dL_dc = denom2inv * (-a * a * dLc3 + 2 * a * b * dLc1 + (denom - a * c) * dLc0);
            dL_db = denom2inv * 2 * (b * c * dLc0 - (denom + 2 * b * b) * dLc1 + a * b * dLc3);
            dcov[0] = (T_(0, 0) * T_(0, 0) * dL_da + T_(0, 0) * T_(1, 0) * dL_db + T_(1, 0) * T_(1, 0) * dL_dc);
            dcov[3] = (T_(0, 1) * T_(0, 1) * dL_da + T_(0, 1) * T_(1, 1) * dL_db + T_(1, 1) * T_(1, 1) * dL_dc);
            dcov[5] = (T_(0, 2) * T_(0, 2) * dL_da + T_(0, 2) * T_(1, 2) * dL_db + T_(1, 2) * T_(1, 2) * dL_dc);
            dcov[1] = 2 * T_(0, 0) * T_(0, 1) * dL_da + (T_(0, 0) * T_(1, 1) + T_(0, 1) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 1) * dL_dc;
            dcov[2] = 2 * T_(0, 0) * T_(0, 2) * dL_da + (T_(0, 0) * T_(1, 2) + T_(0, 2) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 2) * dL_dc;
            dcov[4] = 2 * T_(0, 2) * T_(0, 1) * dL_da + (T_(0, 1) * T_(1, 2) + T_(0, 2) * T_(1, 1)) * dL_db + 2 * T_(1, 1) * T_(1, 2) * dL_dc;
        }
        const float V[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
#define TV_(rw, k) (T_(rw, 0) * V[k][0] + T_(rw, 1) * V[k][1] + T_(rw, 2) * V[k][2])
        const float dL_dT00 = 2 * TV_(0, 0) * dL_da + TV_(1, 0) * dL_db;
        const float dL_dT01 = 2 * TV_(0, 1) * dL_da + TV_(1, 1) * dL_db;
        const float dL_dT02 = 2 * TV_(0, 2) * dL_da + TV_(1, 2) * dL_db;
        const float dL_dT10 = 2 * TV_(1, 0) * dL_dc + TV_(0, 0) * dL_db;
        const float dL_dT11 = 2 * TV_(1, 1) * dL_dc + TV_(0, 1) * dL_db;
        const float dL_dT12 = 2 * TV_(1, 2) * dL_dc + TV_(0, 2) * dL_db;
#undef TV_
#undef T_
#define W_(k, rr) cm.view[4 * (rr) + (k)]
        const float dL_dJ00 = W_(0, 0) * dL_dT00 + W_(0, 1) * dL_dT01 + W_(0, 2) * dL_dT02;
        const float dL_dJ02 = W_(2, 0) * dL_dT00 + W_(2, 1) * dL_dT01 + W_(2, 2) * dL_dT02;
        const float dL_dJ11 = W_(1, 0) * dL_dT10 + W_(1, 1) * dL_dT11 + W_(1, 2) * dL_dT12;
        const float dL_dJ12 = W_(2, 0) * dL_dT10 + W_(2, 1) * dL_dT11 + W_(2, 2) * dL_dT12;
#undef W_
        const float h_x = cm.focal_x, h_y = cm.focal_y;
        const float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dL_dtx = e.xmul * -h_x * tz2 * dL_dJ02;
        const float dL_dty = e.ymul * -h_y * tz2 * dL_dJ12;
        const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * e.t[0]) * tz3 * dL_dJ02 + (2 * h_y * e.t[1]) * tz3 * dL_dJ12;
        const float* vm = cm.view;
        dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
    }
    // ---- projection path (backward.cu:367-387) ----
    {
        const float* proj = cm.proj;
        const float4 m_hom = xform44(mean, proj);
        const float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        const float g2x = part[3], g2y = part[4];
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    }
    // ---- SH path (backward.cu:20-139) ----
    {
        const float dRGB[3] = {part[0], part[1], part[2]};       // (masked by the view's clamp flags where it was packed)
        float wgt[16];
#pragma unroll
        for (int i = 0; i < 16; i++) wgt[i] = 0.0f;
        const int deg = cm.D;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        wgt[0] = kSH0;
        if (deg > 0) { wgt[1] = -kSH1 * y; wgt[2] = kSH1 * z; wgt[3] = -kSH1 * x; }
        if (deg > 1) {
            wgt[4] = kSH2[0] * xy; wgt[5] = kSH2[1] * yz; wgt[6] = kSH2[2] * (2.f * zz - xx - yy);
            wgt[7] = kSH2[3] * xz; wgt[8] = kSH2[4] * (xx - yy);
        }
        if (deg > 2) {
            wgt[9] = kSH3[0] * y * (3.f * xx - yy); wgt[10] = kSH3[1] * xy * z;
            wgt[11] = kSH3[2] * y * (4.f * zz - xx - yy); wgt[12] = kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
            wgt[13] = kSH3[4] * x * (4.f * zz - xx - yy); wgt[14] = kSH3[5] * z * (xx - yy);
            wgt[15] = kSH3[6] * x * (xx - 3.f * yy);
        }
        const float dd0 = shd[0] * dRGB[0] + shd[1] * dRGB[1] + shd[2] * dRGB[2];
        const float dd1 = shd[3] * dRGB[0] + shd[4] * dRGB[1] + shd[5] * dRGB[2];
        const float dd2 = shd[6] * dRGB[0] + shd[7] * dRGB[1] + shd[8] * dRGB[2];
        const float sum2 = dox * dox + doy * doy + doz * doz;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean[0] += ((+sum2 - dox * dox) * dd0 - doy * dox * dd1 - doz * dox * dd2) * invsum32;
        dmean[1] += (-dox * doy * dd0 + (sum2 - doy * doy) * dd1 - doz * doy * dd2) * invsum32;
        dmean[2] += (-dox * doz * dd0 - doy * doz * dd1 + (sum2 - doz * doz) * dd2) * invsum32;
#pragma unroll
        for (int i = 0; i < 48; i++) a_sh[i] += wgt[i / 3] * dRGB[i % 3];
    }
    a_mean[0] += dmean[0]; a_mean[1] += dmean[1]; a_mean[2] += dmean[2];
    a_opac += raw_opacity ? part[8] * ((1.0f - o) * o) : part[8];
    // ---- cov3D -> scale, quaternion (backward.cu:278-341) ----
    {
        const float r = q.x, qx = q.y, qy = q.z, qz = q.w;
        const Rot3 R = quat_to_rot(q);
        const float s[3] = {cm.scale_modifier * sc.x, cm.scale_modifier * sc.y, cm.scale_modifier * sc.z};
        float Mm[3][3];
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++) Mm[cc][rr] = s[rr] * R.c[cc][rr];
        const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
        float dMt[3][3];
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
                dMt[rr][cc] = (Mm[0][rr] * 2.0f) * dS[cc][0] + (Mm[1][rr] * 2.0f) * dS[cc][1] + (Mm[2][rr] * 2.0f) * dS[cc][2];
        float ds[3], dq[4];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) ds[cc] = R.c[0][cc] * dMt[cc][0] + R.c[1][cc] * dMt[cc][1] + R.c[2][cc] * dMt[cc][2];
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++) dMt[cc][rr] *= s[cc];
        dq[0] = 2 * qz * (dMt[0][1] - dMt[1][0]) + 2 * qy * (dMt[2][0] - dMt[0][2]) + 2 * qx * (dMt[1][2] - dMt[2][1]);
        dq[1] = 2 * qy * (dMt[1][0] + dMt[0][1]) + 2 * qz * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * qx * (dMt[2][2] + dMt[1][1]);
        dq[2] = 2 * qx * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * qz * (dMt[1][2] + dMt[2][1]) - 4 * qy * (dMt[2][2] + dMt[0][0]);
        dq[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * qx * (dMt[2][0] + dMt[0][2]) + 2 * qy * (dMt[1][2] + dMt[2][1]) - 4 * qz * (dMt[1][1] + dMt[0][0]);
        if (raw_scale) { ds[0] *= sc.x; ds[1] *= sc.y; ds[2] *= sc.z; }          // d exp = exp
        if (raw_rot) {            // y = x / max(|x|, eps): dx = (g - y (y . g)) / max(|x|, eps)
            const float4 xr = q_raw;
            const float nrm = sqrtf(xr.x * xr.x + xr.y * xr.y + xr.z * xr.z + xr.w * xr.w);
            const float inv = 1.0f / fmaxf(nrm, 1e-12f);
            const float4 yn = make_float4(xr.x * inv, xr.y * inv, xr.z * inv, xr.w * inv);
            const float d = nrm > 1e-12f ? (yn.x * dq[0] + yn.y * dq[1] + yn.z * dq[2] + yn.w * dq[3]) : 0.0f;
            dq[0] = (dq[0] - yn.x * d) * inv; dq[1] = (dq[1] - yn.y * d) * inv;
            dq[2] = (dq[2] - yn.z * d) * inv; dq[3] = (dq[3] - yn.w * d) * inv;
        }
        a_scale[0] += ds[0]; a_scale[1] += ds[1]; a_scale[2] += ds[2];
        a_rot[0] += dq[0]; a_rot[1] += dq[1]; a_rot[2] += dq[2]; a_rot[3] += dq[3];
    }
}

// One workgroup per tile of CMB_TILE Gaussians.  Pass A: per block of 64 Gaussians the views' mask words -> a byte of view
// bits per Gaussian; Gaussians without a row anywhere get their zero rows at once, the others are compacted into an LDS list.
// Pass B: one lane per listed Gaussian walks its views in view order; 59 floats accumulated in registers, written once.
__global__ void __launch_bounds__(CMB_THREADS)
backward_combine_kernel(int first, int n, int n_views, const uint32_t* __restrict__ packets, size_t packet_stride_words,
                        const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ scales,
                        const float* __restrict__ rotations, const float* __restrict__ opacities, RawInputs raw,
                        float* __restrict__ dL_dmean3D, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
                        float* __restrict__ dL_dopacity, float* __restrict__ dL_dsh, uint32_t* __restrict__ status, uint32_t seq,
                        unsigned char* __restrict__ row_live)
{
    __shared__ CmbCam cams[CMB_MAX_VIEWS];
    __shared__ uint32_t list[CMB_TILE];
    __shared__ uint32_t n_list;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t bases_w = sum_packet_bases_word((size_t)n), rows_w = sum_packet_rows_word((size_t)n);
    if (tid < n_views) {
        const uint32_t* h = packets + (size_t)tid * packet_stride_words;
        const float* f = reinterpret_cast<const float*>(h);
        CmbCam& cm = cams[tid];
        for (int i = 0; i < 16; i++) { cm.view[i] = f[8 + i]; cm.proj[i] = f[24 + i]; }
        cm.campos[0] = f[40]; cm.campos[1] = f[41]; cm.campos[2] = f[42];
        cm.tan_fovx = f[43]; cm.tan_fovy = f[44];
        const int W = (int)h[45], H = (int)h[46];
        cm.focal_y = H / (2.0f * cm.tan_fovy);     // api.hip make_view (rasterizer_impl.cu:222-223)
        cm.focal_x = W / (2.0f * cm.tan_fovx);
        cm.half_w = 0.5f * W; cm.half_h = 0.5f * H;
        cm.scale_modifier = f[47]; cm.D = (int)h[48];
    }
    if (tid == 0) n_list = 0u;
    // the exchange's verdict for the host (pinned memory, polled): per view the rows it wanted, then the sequence number
    if (status && blockIdx.x == 0 && tid == 0) {
        uint32_t over = 0;
        for (int v = 0; v < n_views; v++) {
            const uint32_t* h = packets + (size_t)v * packet_stride_words;
            const uint32_t want = h[1];
            __hip_atomic_store(&status[2 + v], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            over |= (want > h[3] || h[5] != FRG_SUM_MAGIC || h[2] != (uint32_t)n || h[4] != (uint32_t)first) ? 1u : 0u;
        }
        __hip_atomic_store(&status[1], over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&status[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    const int tile0 = blockIdx.x * CMB_TILE;        // relative to `first`
    // ---- pass A ----
    for (int bb = wave; bb < CMB_TILE / 64; bb += CMB_THREADS / 64) {
        const int g0 = tile0 + bb * 64;
        if (g0 >= n) break;
        const int blk = g0 / 64, g = g0 + lane;
        const bool valid = g < n;
        uint32_t vb = 0;
        for (int v = 0; v < n_views; v++) {
            const unsigned long long m = reinterpret_cast<const unsigned long long*>(packets + (size_t)v * packet_stride_words + FRG_SUM_HDR_WORDS)[blk];
            vb |= (uint32_t)((m >> lane) & 1ull) << v;
        }
        const bool live = valid && vb != 0u;
        const unsigned long long lm = __builtin_amdgcn_ballot_w64(live);
        uint32_t at = 0;
        if (lane == 0 && lm) at = atomicAdd(&n_list, (uint32_t)__popcll(lm));
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (live) list[at + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))] = (uint32_t)(bb * 64 + lane) | (vb << 16);
        const size_t gi = (size_t)first + g;
        if (row_live && valid) row_live[gi] = live ? 1 : 0;
        if (valid && !live && !row_live) {
            dL_dmean3D[3 * gi] = 0.f; dL_dmean3D[3 * gi + 1] = 0.f; dL_dmean3D[3 * gi + 2] = 0.f;
            dL_dscale[3 * gi] = 0.f; dL_dscale[3 * gi + 1] = 0.f; dL_dscale[3 * gi + 2] = 0.f;
            *reinterpret_cast<float4*>(dL_drot + 4 * gi) = make_float4(0.f, 0.f, 0.f, 0.f);
            dL_dopacity[gi] = 0.f;
        }
        if (!row_live) {
            // the SH rows of the Gaussians without a row in any view: zeros, as one float4 stream over the block
            typedef float nt_f4 __attribute__((ext_vector_type(4)));
            nt_f4* dst = reinterpret_cast<nt_f4*>(dL_dsh) + ((size_t)first + g0) * 12;
            const int nvalid = min(64, n - g0);
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const int f = k * 64 + lane, gl = f / 12;
                if (gl < nvalid && !((lm >> gl) & 1ull)) __builtin_nontemporal_store(nt_f4{0.f, 0.f, 0.f, 0.f}, dst + f);
            }
        }
    }
    __syncthreads();
    // ---- pass B ----
    const uint32_t L = n_list;
    for (uint32_t e0 = 0; e0 < L; e0 += CMB_THREADS) {
        const uint32_t ei = e0 + (uint32_t)tid;
        const bool have = ei < L;
        const uint32_t ent = have ? list[ei] : 0u;
        uint32_t vb = ent >> 16;
        const int gl = (int)(ent & 0xFFFFu), g = tile0 + gl;
        const int idx = first + g, blk = g / 64, gl64 = g & 63;
        float a_mean[3] = {0.f, 0.f, 0.f}, a_scale[3] = {0.f, 0.f, 0.f}, a_rot[4] = {0.f, 0.f, 0.f, 0.f}, a_opac = 0.f, a_sh[48];
#pragma unroll
        for (int i = 0; i < 48; i++) a_sh[i] = 0.f;
        float3 mean = make_float3(0.f, 0.f, 0.f), sc = make_float3(1.f, 1.f, 1.f);
        float4 q = make_float4(1.f, 0.f, 0.f, 0.f), q_raw = q;
        float o = 0.f;
        const float4* sh_row = reinterpret_cast<const float4*>(shs) + (size_t)idx * 12;
        if (have) {
            mean = param_mean(means3D, raw, idx);
            sc = param_scale(scales, raw, idx);
            q = param_rot(rotations, raw, idx);
            o = param_opacity(opacities, raw, idx);
            if (raw.raw_rot) q_raw = make_float4(raw.raw_rot[4 * idx], raw.raw_rot[4 * idx + 1], raw.raw_rot[4 * idx + 2], raw.raw_rot[4 * idx + 3]);
        }
        while (__builtin_amdgcn_ballot_w64(vb != 0u)) {          // wave-uniform trip count: the most views any lane has left
            if (vb != 0u) {
                const int v = __builtin_ctz(vb);
                vb &= vb - 1u;
                const uint32_t* pk = packets + (size_t)v * packet_stride_words;
                const unsigned long long m = reinterpret_cast<const unsigned long long*>(pk + FRG_SUM_HDR_WORDS)[blk];
                const uint32_t row = (pk + bases_w)[blk] + (uint32_t)__popcll(m & ((1ull << gl64) - 1ull));
                if (row < pk[3]) {                                  // (beyond the capacity: the step is repeated, status says so)
                    const float* r = reinterpret_cast<const float*>(pk) + rows_w + (size_t)row * FRG_SUM_ROW_FLOATS;
                    float part[FRG_SLOT_FLOATS];
#pragma unroll
                    for (int c2 = 0; c2 < FRG_SLOT_FLOATS; c2++) part[c2] = r[c2];
                    combine_one_view(cams[v], mean, sc, q, o, raw.raw_opacity != nullptr, raw.raw_scale != nullptr, raw.raw_rot != nullptr,
                                     q_raw, sh_row, part, a_mean, a_scale, a_rot, a_opac, a_sh);
                }
            }
        }
        if (have) {
            const float ds[3] = {a_scale[0], a_scale[1], a_scale[2]}, dq[4] = {a_rot[0], a_rot[1], a_rot[2], a_rot[3]};
            dL_dmean3D[3 * (size_t)idx] = a_mean[0]; dL_dmean3D[3 * (size_t)idx + 1] = a_mean[1]; dL_dmean3D[3 * (size_t)idx + 2] = a_mean[2];
            dL_dscale[3 * (size_t)idx] = ds[0]; dL_dscale[3 * (size_t)idx + 1] = ds[1]; dL_dscale[3 * (size_t)idx + 2] = ds[2];
            *reinterpret_cast<float4*>(dL_drot + 4 * (size_t)idx) = make_float4(dq[0], dq[1], dq[2], dq[3]);
            dL_dopacity[idx] = a_opac;
            typedef float nt_f4 __attribute__((ext_vector_type(4)));
            nt_f4* dst = reinterpret_cast<nt_f4*>(dL_dsh) + (size_t)idx * 12;
#pragma unroll
            for (int j = 0; j < 12; j++)
                __builtin_nontemporal_store(nt_f4{a_sh[4 * j], a_sh[4 * j + 1], a_sh[4 * j + 2], a_sh[4 * j + 3]}, dst + j);
        }
    }
}

hipError_t launch_backward_combine(int first, int n, int n_views, const void* packets, size_t packet_stride_bytes,
                                   const FwdInputs& in, const BwdOutputs& out, uint32_t* status, uint32_t seq, unsigned char* row_live,
                                   hipStream_t s)
{
    const int tiles = (n + CMB_TILE - 1) / CMB_TILE;
    hipLaunchKernelGGL(backward_combine_kernel, dim3(tiles), dim3(CMB_THREADS), 0, s, first, n, n_views,
                       reinterpret_cast<const uint32_t*>(packets), packet_stride_bytes / 4, in.means3D, in.shs, in.scales, in.rotations,
                       in.opacities, in.raw, out.dL_dmean3D, out.dL_dscale, out.dL_drot, out.dL_dopacity, out.dL_dsh, status, seq, row_live);
    return hipGetLastError();
}

}  // namespace frg
