// Slot-sum exchange of the view-parallel step (no counterpart in the single-GPU reference).
//
// After phase 1 of the backward (blend backward + slot reduction) everything one view contributes to the gradient of a
// Gaussian is determined by TWELVE floats -- the clamp-masked colour gradient, the six pixel moments of G dL/dalpha
// (preprocess_bwd.hip, "sums") and the three view-direction terms d(colour)/d(direction) . dRGB (what phase 2 forms from the
// forward's sh_dir) -- together with data every rank already holds: the parameters and the view's camera.
// And only the Gaussians some pixel reached before its tile saturated have sums at all (one in eight at C3).  So the
// ranks exchange those sums instead of finished gradients:
//
//   pack     the rows {dRGB[3], moments[6], dd[3]} (48 bytes) of the Gaussians with a gradient, IN INDEX ORDER, behind a bit mask
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
// forward's own functions (gauss_math.h), bit-identically.  The view-direction terms travel in the rows because forming them
// needs the 192-byte SH row of the Gaussian: read once per (Gaussian, view) pair by every rank, those rows were the larger
// part of the combine pass's memory traffic (0.84 GB fetched for eight C3 views; 12 more bytes per row on the wire instead).
#include "gauss_math.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace frg {

// ---- packet layout (uint32 words) -------------------------------------------------------------------------------------------
//   [0] rows packed  [1] rows wanted (> capacity: overflow)  [2] Gaussians of the packet  [3] capacity  [4] first Gaussian
//   [5] magic  [8..23] viewmatrix  [24..39] projmatrix  [40..42] camera centre  [43] tan_fovx  [44] tan_fovy
//   [45] width  [46] height  [47] scale_modifier  [48] active SH degree  [49] focal_x  [50] focal_y (api.hip make_view's)
//   masks   uint64[nblk]  at word 64                 (nblk = blocks of 64 Gaussians)
//   bases   uint32[nblk]  behind them, 16-byte aligned
//   rows    float[capacity][12] behind them, 16-byte aligned
__host__ __device__ inline size_t sum_packet_blocks(size_t n) { return (n + 63) / 64; }
__host__ __device__ inline size_t sum_packet_bases_word(size_t n) { return FRG_SUM_HDR_WORDS + 2 * sum_packet_blocks(n); }
__host__ __device__ inline size_t sum_packet_rows_word(size_t n) { return (sum_packet_bases_word(n) + sum_packet_blocks(n) + 3) / 4 * 4; }
size_t sum_packet_bytes(size_t n, size_t capacity) { return ((sum_packet_rows_word(n) + FRG_SUM_ROW_FLOATS * capacity + 3) / 4 * 4) * 4; }

// Pack, two launches over the packet's blocks of 64 Gaussians:
//   sum_rows_local_kernel   one thread per block, 256 blocks per workgroup: the mask word (copied from the phase-1 workspace),
//                           its popcount, the exclusive prefix WITHIN the group of 256 blocks -> bases[], the group's total ->
//                           group_tot[] (scratch behind the masks in the workspace);
//   sum_rows_pack_kernel    one wave per block: the prefix of the groups before its own (a few hundred totals: three loads
//                           per lane and a wave scan) makes bases[] global; the rows of the marked Gaussians, in index order;
//                           the first wave writes the header (rows wanted = the sum of all totals, the camera).
// (Round 6's first forms scanned all blocks in ONE workgroup: 70, 30 and again 70 us per packet of 47 000 blocks -- one CU's
// load path.)
#define SUM_GROUP 256
__global__ void __launch_bounds__(SUM_GROUP)
sum_rows_local_kernel(int first, int n, const unsigned long long* __restrict__ live_masks, uint32_t* __restrict__ packet,
                      uint32_t* __restrict__ group_tot)
{
    __shared__ uint32_t wave_tot[SUM_GROUP / 64];
    const int nblk = (int)sum_packet_blocks((size_t)n), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x * SUM_GROUP + tid;
    const unsigned long long m = b < nblk ? (live_masks + first / 64)[b] : 0ull;
    const uint32_t c = (uint32_t)__popcll(m), incl = wave_incl_scan_dpp(c);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SUM_GROUP / 64; w++) { const uint32_t t = wave_tot[w]; if (w < wave) before += t; total += t; }
    if (b < nblk) {
        reinterpret_cast<unsigned long long*>(packet + FRG_SUM_HDR_WORDS)[b] = m;
        (packet + sum_packet_bases_word((size_t)n))[b] = before + incl - c;
    }
    if (tid == 0) group_tot[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256)
sum_rows_pack_kernel(int first, int n, uint32_t capacity, const float* __restrict__ sums, const float* __restrict__ view_dir_terms,
                     const float* __restrict__ drgb_masked, uint32_t* __restrict__ packet, const uint32_t* __restrict__ group_tot, SumCamera cam,
                     const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos)
{
    const int lane = threadIdx.x & 63, blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nblk = (int)sum_packet_blocks((size_t)n);
    if (blk >= nblk) return;
    const int ngroups = (nblk + SUM_GROUP - 1) / SUM_GROUP, grp = blk / SUM_GROUP;
    // rows before this block's group (and, for the header, of all groups)
    uint32_t before = 0, all = 0;
    for (int j = lane; j < ngroups; j += 64) { const uint32_t t = group_tot[j]; all += t; if (j < grp) before += t; }
    before = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(before), 63);
    uint32_t* bases = packet + sum_packet_bases_word((size_t)n);
    const uint32_t base = before + bases[blk];
    if (blk == 0) {
        all = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(all), 63);
        float* f = reinterpret_cast<float*>(packet);
        if (lane < 16) { f[8 + lane] = viewmatrix[lane]; f[24 + lane] = projmatrix[lane]; }
        if (lane < 3) f[40 + lane] = campos[lane];
        if (lane >= 51 && lane < FRG_SUM_HDR_WORDS) packet[lane] = 0u;
        if (lane == 0) {
            packet[0] = all < capacity ? all : capacity;
            packet[1] = all;
            packet[2] = (uint32_t)n; packet[3] = capacity; packet[4] = (uint32_t)first; packet[5] = FRG_SUM_MAGIC;
            packet[6] = 0u; packet[7] = 0u;
            f[43] = cam.tan_fovx; f[44] = cam.tan_fovy;
            packet[45] = (uint32_t)cam.width; packet[46] = (uint32_t)cam.height;
            f[47] = cam.scale_modifier; packet[48] = (uint32_t)cam.D;
            f[49] = cam.width / (2.0f * cam.tan_fovx);        // rasterizer_impl.cu:222-223, as api.hip make_view forms them
            f[50] = cam.height / (2.0f * cam.tan_fovy);
        }
    }
    const unsigned long long m = reinterpret_cast<const unsigned long long*>(packet + FRG_SUM_HDR_WORDS)[blk];
    if (lane == 0) bases[blk] = base;                   // (this wave alone reads and writes the block's entry)
    if (!((m >> lane) & 1ull)) return;
    const uint32_t row = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (row >= capacity) return;                 // (over capacity: the header says so; the host packs again into a larger packet)
    const size_t g = (size_t)first + (size_t)blk * 64 + lane;
    float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(packet) + sum_packet_rows_word((size_t)n) + (size_t)row * FRG_SUM_ROW_FLOATS);
    const float* sp = sums + g * FRG_SLOT_FLOATS;
    const float* dd = view_dir_terms + 3 * g;
    dst[0] = make_float4(drgb_masked[3 * g], drgb_masked[3 * g + 1], drgb_masked[3 * g + 2], sp[3]);
    dst[1] = make_float4(sp[4], sp[5], sp[6], sp[7]);
    dst[2] = make_float4(sp[8], dd[0], dd[1], dd[2]);
}

hipError_t launch_pack_sum_rows(int first, int n, uint32_t capacity, const unsigned long long* live_masks, const float* sums,
                                const float* view_dir_terms, const float* drgb_masked, const SumCamera& cam, const float* viewmatrix, const float* projmatrix,
                                const float* campos, void* packet, uint32_t* group_tot, hipStream_t s)
{
    uint32_t* pk = reinterpret_cast<uint32_t*>(packet);
    const int nblk = (int)sum_packet_blocks((size_t)n);
    hipLaunchKernelGGL(sum_rows_local_kernel, dim3((nblk + SUM_GROUP - 1) / SUM_GROUP), dim3(SUM_GROUP), 0, s, first, n, live_masks, pk, group_tot);
    hipLaunchKernelGGL(sum_rows_pack_kernel, dim3((nblk + 3) / 4), dim3(256), 0, s, first, n, capacity, sums, view_dir_terms, drgb_masked, pk, group_tot, cam,
                       viewmatrix, projmatrix, campos);
    return hipGetLastError();
}

// ---- combine ------------------------------------------------------------------------------------------------------------------
#define CMB_MAX_VIEWS 16

struct CmbCam { float view[16], proj[16], campos[3], tan_fovx, tan_fovy, focal_x, focal_y, half_w, half_h, scale_modifier; int D; };

// A view's camera from its packet header.  The address is wave-uniform, but behind the passes' stores the compiler does not
// prove these loads unclobbered and issues vector loads: every value is moved to a scalar register by hand (v_readfirstlane),
// so that the camera costs no vector registers across the chain.
__device__ __forceinline__ float uniform_f(const uint32_t* __restrict__ h, int i) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)h[i])); }
__device__ __forceinline__ void load_cam(const uint32_t* __restrict__ h, CmbCam& cm)
{
#pragma unroll
    for (int i = 0; i < 16; i++) { cm.view[i] = uniform_f(h, 8 + i); cm.proj[i] = uniform_f(h, 24 + i); }
    cm.campos[0] = uniform_f(h, 40); cm.campos[1] = uniform_f(h, 41); cm.campos[2] = uniform_f(h, 42);
    cm.tan_fovx = uniform_f(h, 43); cm.tan_fovy = uniform_f(h, 44);
    cm.focal_x = uniform_f(h, 49); cm.focal_y = uniform_f(h, 50);
    cm.half_w = 0.5f * (float)__builtin_amdgcn_readfirstlane((int)h[45]); cm.half_h = 0.5f * (float)__builtin_amdgcn_readfirstlane((int)h[46]);
    cm.scale_modifier = uniform_f(h, 47); cm.D = __builtin_amdgcn_readfirstlane((int)h[48]);
}

// sections 2, 3 and 5 of preprocess_bwd_kernel, and the view-direction term of section 4, for ONE (Gaussian, view): `part` = the
// view's nine slot sums of the Gaussian (the colour part clamp-masked), dd = its three view-direction terms.  Adds the view's
// terms to acc[11].  Expression for expression the chain of preprocess_bwd.hip (has_grad branch); tests pin the two bit for
// bit.  Raw-parameter mode (raw_params.h): the activations' Jacobians are applied per view, as phase 2 applies them.
__device__ __forceinline__ void combine_one_view(const CmbCam& cm, const float3 mean, const float3 sc, const float4 q, const float o,
                                                 const bool raw_opacity, const bool raw_scale, const bool raw_rot, const float4 q_raw,
                                                 float (&part)[FRG_SLOT_FLOATS], const float dd0, const float dd1, const float dd2,
                                                 float* __restrict__ acc /* [11]: mean3D 3, scale 3, rot 4, opacity */)
{
    const float dox = mean.x - cm.campos[0], doy = mean.y - cm.campos[1], doz = mean.z - cm.campos[2];
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
    // ---- view-direction term of the SH path (backward.cu:130-138; auxiliary.h:107-117 dnormvdv) ----
    {
        const float sum2 = dox * dox + doy * doy + doz * doz;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean[0] += ((+sum2 - dox * dox) * dd0 - doy * dox * dd1 - doz * dox * dd2) * invsum32;
        dmean[1] += (-dox * doy * dd0 + (sum2 - doy * doy) * dd1 - doz * doy * dd2) * invsum32;
        dmean[2] += (-dox * doz * dd0 - doy * doz * dd1 + (sum2 - doz * doz) * dd2) * invsum32;
    }
    acc[0] += dmean[0]; acc[1] += dmean[1]; acc[2] += dmean[2];
    acc[10] += raw_opacity ? part[8] * ((1.0f - o) * o) : part[8];
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
        acc[3] += ds[0]; acc[4] += ds[1]; acc[5] += ds[2];
        acc[6] += dq[0]; acc[7] += dq[1]; acc[8] += dq[2]; acc[9] += dq[3];
    }
}

// The exchange's verdict for the host (pinned memory, polled): word 0 = any packet overflowed / does not describe this range,
// words 1 .. n_views = the rows each view wanted -- every one a 64-bit (sequence number << 32 | value) stored at once, so the
// host knows a word is this pass's by its tag and no fence is needed (a system-scope release here writes the L2's dirty
// lines back first: the one-thread kernel took 60 - 160 us behind a backward).  Its own launch in front of the passes: the host
// learns the verdict as early as it can be known, and the passes keep their scalar loads.
__global__ void combine_verdict_kernel(unsigned long long* __restrict__ status, uint32_t seq, const uint32_t* __restrict__ packets,
                                       size_t packet_stride_words, int n_views, int first, int n)
{
    const int v = threadIdx.x;
    uint32_t bad = 0;
    if (v < n_views) {
        const uint32_t* h = packets + (size_t)v * packet_stride_words;
        const uint32_t want = h[1];
        bad = (want > h[3] || h[5] != FRG_SUM_MAGIC || h[2] != (uint32_t)n || h[4] != (uint32_t)first) ? 1u : 0u;
        __hip_atomic_store(&status[1 + v], ((unsigned long long)seq << 32) | want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned long long any = __builtin_amdgcn_ballot_w64(bad != 0u);
    if (v == 0) __hip_atomic_store(&status[0], ((unsigned long long)seq << 32) | (any ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The dense part in three launches over a scratch of 48 bytes per packed row (n_views x capacity_rows rows):
//   combine_index_kernel   one lane per Gaussian, its views in a wave-uniform loop: gidx[v][row] <- the Gaussian of row `row` of
//                          view v (the packets give Gaussian -> row; the chain wants row -> Gaussian);
//   combine_chain_kernel   grid (views, rows / 256): ONE LANE PER PACKED ROW -- every lane busy, the view uniform per workgroup
//                          (its camera in scalar registers), every load of a pair in flight at once, no barrier anywhere;
//                          the pair's eleven terms go to stage[v][row];
//   combine_sum_kernel     one lane per Gaussian: the staged terms of its views added IN VIEW ORDER, every row written once
//                          (zeros where no view has a row).
// Forms of round 6 that were measured and replaced (eight C3 views, 3 M Gaussians, 3 M pairs): one lane per Gaussian walking
// its views, accumulators in registers -- 27 % busy lanes: 0.76 ms; tiles of 1024 / 512 Gaussians with the views in an outer
// loop and the sums in LDS -- dense lanes, but a tile's eight view steps are a chain of barriers with one or two waves at
// work: 0.41 / 0.44 ms.
#define CMB_ACC 11
#define CMB_STAGE 12                    // floats per staged row (48 bytes: three aligned float4)
__global__ void __launch_bounds__(256)
combine_index_kernel(int n, int n_views, const uint32_t* __restrict__ packets, size_t packet_stride_words, uint32_t capacity,
                     uint32_t* __restrict__ gidx)
{
    const int lane = threadIdx.x & 63, blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= (int)sum_packet_blocks((size_t)n)) return;
    const size_t bases_w = sum_packet_bases_word((size_t)n);
    for (int v = 0; v < n_views; v++) {
        const uint32_t* pk = packets + (size_t)v * packet_stride_words;
        const unsigned long long m = reinterpret_cast<const unsigned long long*>(pk + FRG_SUM_HDR_WORDS)[blk];
        if (!((m >> lane) & 1ull)) continue;
        const uint32_t row = (pk + bases_w)[blk] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (row < capacity) gidx[(size_t)v * capacity + row] = (uint32_t)(blk * 64 + lane);
    }
}

__global__ void __launch_bounds__(256)
combine_chain_kernel(int first, int n, const uint32_t* __restrict__ packets, size_t packet_stride_words, uint32_t capacity,
                     const uint32_t* __restrict__ gidx, float* __restrict__ stage,
                     const float* __restrict__ means3D, const float* __restrict__ scales,
                     const float* __restrict__ rotations, const float* __restrict__ opacities, RawInputs raw)
{
    // the VIEW is the fast grid dimension: the workgroups of one row block in all views are dispatched together, and row block k
    // of every view covers about the same Gaussians (the rows are index-ordered, the views' densities alike) -- their
    // parameter lines and SH rows are then found in the L2 by all but the first view
    const int v = blockIdx.x;
    const uint32_t* pk = packets + (size_t)v * packet_stride_words;
    const uint32_t rows = min((uint32_t)__builtin_amdgcn_readfirstlane((int)pk[1]), capacity);
    if (blockIdx.y * 256u >= rows) return;                    // workgroup-uniform
    CmbCam cm;
    load_cam(pk, cm);
    const uint32_t row = blockIdx.y * 256u + threadIdx.x;
    if (row >= rows) return;
    const int idx = first + (int)gidx[(size_t)v * capacity + row];
    const float4* r = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(pk) + sum_packet_rows_word((size_t)n) + (size_t)row * FRG_SUM_ROW_FLOATS);
    const float4 r0 = r[0], r1 = r[1], r2 = r[2];
    float part[FRG_SLOT_FLOATS] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x};
    const float3 mean = param_mean(means3D, raw, idx);
    const float3 sc = param_scale(scales, raw, idx);
    const float4 q = param_rot(rotations, raw, idx);
    const float o = param_opacity(opacities, raw, idx);
    float4 q_raw = q;
    if (raw.raw_rot) q_raw = make_float4(raw.raw_rot[4 * idx], raw.raw_rot[4 * idx + 1], raw.raw_rot[4 * idx + 2], raw.raw_rot[4 * idx + 3]);
    float out[CMB_ACC];
#pragma unroll
    for (int k = 0; k < CMB_ACC; k++) out[k] = 0.0f;
    combine_one_view(cm, mean, sc, q, o, raw.raw_opacity != nullptr, raw.raw_scale != nullptr, raw.raw_rot != nullptr, q_raw,
                     part, r2.y, r2.z, r2.w, out);
    float4* dst = reinterpret_cast<float4*>(stage + ((size_t)v * capacity + row) * CMB_STAGE);
    dst[0] = make_float4(out[0], out[1], out[2], out[3]);
    dst[1] = make_float4(out[4], out[5], out[6], out[7]);
    dst[2] = make_float4(out[8], out[9], out[10], 0.0f);
}

// The two gather passes, one lane per Gaussian each, over the views in which it has a row, IN VIEW ORDER:
//   combine_sum_kernel   the 11 dense sums: the terms the chain pass staged for (view, row), added one after the other;
//   combine_sh_kernel    dL_dsh: basis(dir_v) (x) dRGB_v, 48 products per view added into 48 accumulators; the 192-byte rows
//                        leave through a wave-private LDS transpose as contiguous float4 streams (view_exchange.hip's idiom);
// every row of the five outputs written once, zeros where no view has a row (with row_live: those are skipped and the byte
// says so).  The view loops are wave-uniform (mask word, row base and camera centre are uniform loads), in groups of four
// views whose look-ups are requested together.  (As ONE kernel the 59 accumulators and the staged rows of a group took 180
// VGPRs, two waves per SIMD: 0.36 ms for eight C3 views against 0.10 + 0.19 of the two.)
__global__ void __launch_bounds__(256)
combine_sum_kernel(int first, int n, int n_views, const uint32_t* __restrict__ packets, size_t packet_stride_words, uint32_t capacity,
                   const float* __restrict__ stage, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dscale,
                   float* __restrict__ dL_drot, float* __restrict__ dL_dopacity, unsigned char* __restrict__ row_live)
{
    const int lane = threadIdx.x & 63, blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= (int)sum_packet_blocks((size_t)n)) return;
    const int g = blk * 64 + lane;
    const size_t bases_w = sum_packet_bases_word((size_t)n);
    float a[CMB_ACC];
#pragma unroll
    for (int k = 0; k < CMB_ACC; k++) a[k] = 0.0f;
    bool live = false;
#pragma unroll 1
    for (int v0 = 0; v0 < n_views; v0 += 4) {
        unsigned long long m[4];
        uint32_t base[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int v = min(v0 + u, n_views - 1);
            const uint32_t* pk = packets + (size_t)v * packet_stride_words;
            m[u] = v0 + u < n_views ? reinterpret_cast<const unsigned long long*>(pk + FRG_SUM_HDR_WORDS)[blk] : 0ull;
            base[u] = (pk + bases_w)[blk];
        }
        float4 st[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int v = min(v0 + u, n_views - 1);
            const bool bit = ((m[u] >> lane) & 1ull) != 0ull;
            live |= bit;
            const uint32_t row = base[u] + (uint32_t)__popcll(m[u] & ((1ull << lane) - 1ull));
            st[u][0] = st[u][1] = st[u][2] = make_float4(0.f, 0.f, 0.f, 0.f);
            m[u] = (bit && row < capacity) ? 1ull : 0ull;
            if (m[u]) {
                const float4* sp = reinterpret_cast<const float4*>(stage + ((size_t)v * capacity + row) * CMB_STAGE);
                st[u][0] = sp[0]; st[u][1] = sp[1]; st[u][2] = sp[2];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {                          // view order
            if (m[u]) {
                a[0] += st[u][0].x; a[1] += st[u][0].y; a[2] += st[u][0].z; a[3] += st[u][0].w;
                a[4] += st[u][1].x; a[5] += st[u][1].y; a[6] += st[u][1].z; a[7] += st[u][1].w;
                a[8] += st[u][2].x; a[9] += st[u][2].y; a[10] += st[u][2].z;
            }
        }
    }
    if (g >= n) return;
    const size_t gi = (size_t)first + g;
    if (row_live) { row_live[gi] = live ? 1 : 0; if (!live) return; }
    dL_dmean3D[3 * gi] = a[0]; dL_dmean3D[3 * gi + 1] = a[1]; dL_dmean3D[3 * gi + 2] = a[2];
    dL_dscale[3 * gi] = a[3]; dL_dscale[3 * gi + 1] = a[4]; dL_dscale[3 * gi + 2] = a[5];
    *reinterpret_cast<float4*>(dL_drot + 4 * gi) = make_float4(a[6], a[7], a[8], a[9]);
    dL_dopacity[gi] = a[10];
}

#define CSH_SUB 16
#define CSH_ROW_F4 13
__global__ void __launch_bounds__(256)
combine_sh_kernel(int first, int n, int n_views, const uint32_t* __restrict__ packets, size_t packet_stride_words, uint32_t capacity,
                  const float* __restrict__ means3D, RawInputs raw, float* __restrict__ dL_dsh, const unsigned char* __restrict__ row_live)
{
    __shared__ __attribute__((aligned(16))) float4 lds_all[4 * CSH_SUB * CSH_ROW_F4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* shbuf = lds_all + wave * CSH_SUB * CSH_ROW_F4;
    const int blk = blockIdx.x * 4 + wave, g0 = blk * 64;
    if (g0 >= n) return;
    const int g = g0 + lane, idx = first + g;
    const bool valid = g < n;
    const size_t bases_w = sum_packet_bases_word((size_t)n), rows_w = sum_packet_rows_word((size_t)n);
    float out[48];
#pragma unroll
    for (int i = 0; i < 48; i++) out[i] = 0.0f;
    float3 mean = make_float3(0.f, 0.f, 0.f);
    if (valid) mean = param_mean(means3D, raw, idx);
    unsigned long long any = 0ull;
#pragma unroll 1
    for (int v0 = 0; v0 < n_views; v0 += 4) {
        unsigned long long m[4];
        uint32_t base[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int v = min(v0 + u, n_views - 1);
            const uint32_t* pk = packets + (size_t)v * packet_stride_words;
            m[u] = v0 + u < n_views ? reinterpret_cast<const unsigned long long*>(pk + FRG_SUM_HDR_WORDS)[blk] : 0ull;
            base[u] = (pk + bases_w)[blk];
        }
        float dr[4][3];
        bool has[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int v = min(v0 + u, n_views - 1);
            const uint32_t* pk = packets + (size_t)v * packet_stride_words;
            any |= m[u];
            const uint32_t row = base[u] + (uint32_t)__popcll(m[u] & ((1ull << lane) - 1ull));
            has[u] = ((m[u] >> lane) & 1ull) && row < capacity;
            dr[u][0] = dr[u][1] = dr[u][2] = 0.0f;
            if (has[u]) {
                const float* r = reinterpret_cast<const float*>(pk) + rows_w + (size_t)row * FRG_SUM_ROW_FLOATS;
                dr[u][0] = r[0]; dr[u][1] = r[1]; dr[u][2] = r[2];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!__builtin_amdgcn_ballot_w64(has[u])) continue;            // wave-uniform
            const int v = min(v0 + u, n_views - 1);
            const uint32_t* pk = packets + (size_t)v * packet_stride_words;
            const int D = __builtin_amdgcn_readfirstlane((int)pk[48]);
            const float cx = uniform_f(pk, 40), cy = uniform_f(pk, 41), cz = uniform_f(pk, 42);
            if (has[u]) {
                const float dRGB[3] = {dr[u][0], dr[u][1], dr[u][2]};
                const float dox = mean.x - cx, doy = mean.y - cy, doz = mean.z - cz;
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
        }
    }
    // SH rows to write: all of the block's -- or, with row_live, those of its Gaussians with a row somewhere
    const unsigned long long wmask = row_live ? any : ~0ull;
    float4* dst = reinterpret_cast<float4*>(dL_dsh) + ((size_t)first + g0) * 12;
    const int nvalid = min(64, n - g0);
#pragma unroll 1
    for (int h = 0; h < 64 / CSH_SUB; h++) {
        if (((wmask >> (h * CSH_SUB)) & ((1ull << CSH_SUB) - 1ull)) == 0ull) continue;
        if ((lane / CSH_SUB) == h) {
#pragma unroll
            for (int j = 0; j < 12; j++)
                shbuf[(lane % CSH_SUB) * CSH_ROW_F4 + j] = make_float4(out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < CSH_SUB * 12 / 64; k++) {
            const int f = k * 64 + lane, gl = f / 12, j = f - gl * 12;
            if (h * CSH_SUB + gl < nvalid && ((wmask >> (h * CSH_SUB + gl)) & 1ull)) {
                typedef float nt_f4 __attribute__((ext_vector_type(4)));
                const float4 v4 = shbuf[gl * CSH_ROW_F4 + j];
                __builtin_nontemporal_store(nt_f4{v4.x, v4.y, v4.z, v4.w}, reinterpret_cast<nt_f4*>(dst + (size_t)h * CSH_SUB * 12 + f));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

size_t combine_workspace_bytes(int n_views, size_t capacity)
{
    const size_t rows = (size_t)(n_views > 0 ? n_views : 1) * (capacity > 0 ? capacity : 1);
    return align_up(rows * 4, 256) + align_up(rows * CMB_STAGE * 4, 256);
}

hipError_t launch_backward_combine(int first, int n, int n_views, const void* packets, size_t packet_stride_bytes, uint32_t capacity,
                                   const FwdInputs& in, const BwdOutputs& out, unsigned long long* status, uint32_t seq, unsigned char* row_live,
                                   char* workspace, hipStream_t s)
{
    const uint32_t* pk = reinterpret_cast<const uint32_t*>(packets);
    const size_t stride_w = packet_stride_bytes / 4;
    const int nblk = (int)sum_packet_blocks((size_t)n);
    const size_t rows = (size_t)n_views * (capacity > 0 ? capacity : 1);
    uint32_t* gidx = reinterpret_cast<uint32_t*>(workspace);
    float* stage = reinterpret_cast<float*>(workspace + align_up(rows * 4, 256));
    if (status) hipLaunchKernelGGL(combine_verdict_kernel, dim3(1), dim3(64), 0, s, status, seq, pk, stride_w, n_views, first, n);
    hipLaunchKernelGGL(combine_index_kernel, dim3((nblk + 3) / 4), dim3(256), 0, s, n, n_views, pk, stride_w, capacity, gidx);
    if (capacity > 0)
        hipLaunchKernelGGL(combine_chain_kernel, dim3(n_views, (capacity + 255) / 256), dim3(256), 0, s, first, n, pk, stride_w, capacity, gidx, stage,
                           in.means3D, in.scales, in.rotations, in.opacities, in.raw);
    hipLaunchKernelGGL(combine_sum_kernel, dim3((nblk + 3) / 4), dim3(256), 0, s, first, n, n_views, pk, stride_w, capacity, stage,
                       out.dL_dmean3D, out.dL_dscale, out.dL_drot, out.dL_dopacity, row_live);
    // (row_live is written by the sum pass; the SH pass derives the same bits from the masks themselves)
    hipLaunchKernelGGL(combine_sh_kernel, dim3((nblk + 3) / 4), dim3(256), 0, s, first, n, n_views, pk, stride_w, capacity, in.means3D, in.raw,
                       out.dL_dsh, row_live);
    return hipGetLastError();
}

}  // namespace frg
