// Forward per-Gaussian stage, tile counting, scans and the tile-major scatter.
//
// Replaces (behaviour, not structure) the reference's K1-K3/K5:
//   FORWARD::preprocess      forward.cu:155-256   -> preprocess_fwd_kernel
//   cub InclusiveSum         rasterizer_impl.cu:277 -> block sums here + scan_kernel + scatter_kernel
//   duplicateWithKeys        rasterizer_impl.cu:70-111  -> scatter_kernel (tile-major, no 64-bit keys)
//   identifyTileRanges       rasterizer_impl.cu:116-138 -> scan_kernel (ranges from per-tile counts)
// Instead of emitting (tile|depth) 64-bit keys Gaussian-major and radix-sorting
// R of them globally, instances are counted per tile, scattered straight into
// their tile's segment and each segment is depth-sorted in LDS (sort.hip).
#include "gauss_math.h"
#include "kernels.h"

#include <algorithm>

#pragma clang fp contract(off)

namespace frg {

// wave64 inclusive scan (DPP, frg_common.h)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
    (void)lane;
    return wave_incl_scan_dpp(v);
}

// Walk all (Gaussian, tile) instances of the wave's 64 Gaussians with the 64 lanes in
// lock-step: lane l of step k handles instance 64 k + l of the wave's concatenated tile
// rectangles (owner found by binary search over the per-lane starts in LDS).  A per-lane
// `for y, for x` loop instead runs for as long as the LARGEST rectangle in the wave
// (heavy-tailed: ~50 iterations against an average of 6.5 instances per Gaussian).
// lds_start: 65 words, lds_info: 64 int4, both private to the wave.  f(owner lane, tile, tx, ty, payload, payload2).
template <typename F>
__device__ __forceinline__ void wave_for_each_instance(uint32_t touched, int x0, int y0, int rect_w, uint32_t payload,
                                                        uint32_t payload2, uint32_t* lds_start, int4* lds_info, int gx, F&& f)
{
    const int lane = threadIdx.x & 63;
    const uint32_t incl = wave_incl_scan(touched, lane);
    const uint32_t S = (uint32_t)__shfl((int)incl, 63, 64);
    if (S == 0) return;   // wave-uniform
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    lds_start[lane] = incl - touched;
    if (lane == 0) lds_start[64] = S;
    lds_info[lane] = make_int4(x0 | (y0 << 16), rect_w, (int)payload, (int)payload2);   // tile coordinates < 2^16
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t s0 = 0; s0 < S; s0 += 64) {
        const uint32_t s = s0 + lane;
        if (s < S) {
            int owner = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const int mid = owner + step;
                if (mid < 64 && lds_start[mid] <= s) owner = mid;
            }
            const int4 info = lds_info[owner];
            const uint32_t k = s - lds_start[owner];
            uint32_t ry, rx;
            rect_divmod(k, (uint32_t)info.y, ry, rx);
            const int tx = (info.x & 0xFFFF) + (int)rx, ty = (int)((uint32_t)info.x >> 16) + (int)ry;
            f(owner, ty * gx + tx, tx, ty, (uint32_t)info.z, (uint32_t)info.w);
        }
    }
}

// Per-Gaussian forward math (forward.cu:155-256).  Returns tiles_touched; fills the
// three 16-byte records and the tile rectangle when the Gaussian is kept.
__device__ __forceinline__ uint32_t
preprocess_one(int idx, const ViewParams& vp, const ViewMats& vmx,
               const float* __restrict__ means3D, const float* __restrict__ scales,
               const float* __restrict__ rotations, const float* __restrict__ opacities,
               const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
               const float* __restrict__ colors_precomp, const unsigned char* __restrict__ keep_mask, const RawInputs& raw,
               float4* rec /* this Gaussian's record: [0] xydr, [1] conic + opacity */,
               Counters* __restrict__ counters, int prefiltered, int& radius_i, int& x0, int& y0, int& x1, int& y1,
               float3& dir, float& depth)
{
    radius_i = 0;
    if (keep_mask && !keep_mask[idx]) return 0u;   // occlusion-culled by the caller: not part of this view
    const float3 p = param_mean(means3D, raw, idx);
    const float4 p_hom = xform44(p, vmx.proj);
    const float3 p_view = xform43(p, vmx.view);
    const float p_w = 1.0f / (p_hom.w + 0.0000001f);
    const float p_projx = p_hom.x * p_w, p_projy = p_hom.y * p_w;
    if (p_view.z <= 0.2f) {  // near cull only (auxiliary.h:154)
        if (prefiltered) counters->filtered = 1;  // auxiliary.h:156-160: reported by the host instead of __trap()
        return 0u;
    }
    float cov[6];
    if (cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; i++) cov[i] = cov3D_precomp[6 * idx + i];
    } else {
        const float3 s = param_scale(scales, raw, idx);
        const float4 q = param_rot(rotations, raw, idx);
        cov3d_from_scale_rot(s, vp.scale_modifier, q, cov);
    }
    const Ewa e = ewa_setup(p, vp.focal_x, vp.focal_y, vp.tan_fovx, vp.tan_fovy, vmx.view);
    float ca, cb, cc;
    ewa_cov2d(e, cov, ca, cb, cc);
    ca += 0.3f; cc += 0.3f;
    const float det = ca * cc - cb * cb;
    if (det == 0.0f) return 0u;
    const float det_inv = 1.f / det;
    const float mid = 0.5f * (ca + cc);
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    const float px = ndc_to_pix(p_projx, vp.W), py = ndc_to_pix(p_projy, vp.H);
    tile_rect(px, py, f2i(my_radius), vp.gx, vp.gy, x0, y0, x1, y1);
    const uint32_t touched = (uint32_t)(y1 - y0) * (uint32_t)(x1 - x0);
    if (touched == 0) return 0u;
    radius_i = f2i(my_radius);
    depth = p_view.z;
    rec[0] = make_float4(px, py, p_view.z, my_radius);
    rec[1] = make_float4(cc * det_inv, -cb * det_inv, ca * det_inv, param_opacity(opacities, raw, idx));
    if (!colors_precomp) {  // unit view direction for the SH colour (forward.cu:25-27)
        float dx = p.x - vmx.campos[0], dy = p.y - vmx.campos[1], dz = p.z - vmx.campos[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        dir = make_float3(dx / len, dy / len, dz / len);
    }
    return touched;
}

// SH -> RGB in the reference's left-to-right order (forward.cu:28-70), fed one coefficient at a
// time (e = 3 * i + ch) so that the coefficients can arrive in storage order.
struct ShAccum {
    float acc[3];
    __device__ __forceinline__ void add(int i, int ch, float w, float s)
    {
        if (i == 0) acc[ch] = w * s;
        else if (i == 1 || i == 3) acc[ch] = acc[ch] - w * s;
        else acc[ch] = acc[ch] + w * s;
    }
    __device__ __forceinline__ float4 finish()
    {
        uint32_t clamp_bits = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            acc[ch] += 0.5f;
            if (acc[ch] < 0) clamp_bits |= (1u << ch);
            acc[ch] = fmaxf(acc[ch], 0.0f);
        }
        return make_float4(acc[0], acc[1], acc[2], __uint_as_float(clamp_bits));
    }
};

// block-wide inclusive scan of one uint per thread (NW waves); returns the inclusive
// value, *total receives the block sum.  wsum: NW words of LDS.
template <int NW>
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t* wsum, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t x = wsum[w];
        if (w < wave) off += x;
        tot += x;
    }
    *total = tot;
    return off + inc;
}

// A wave of 64 consecutive Gaussians whose instances number more than FRG_BWD_HEAVY_SLOTS goes on the list of the
// per-Gaussian backward's 16-wave launch (GeomState::heavy_waves).  inc: the block-inclusive scan of tiles_touched.
__device__ __forceinline__ void note_heavy_wave(uint32_t inc, uint32_t touched, int first_gaussian, uint32_t* __restrict__ heavy_waves)
{
    const uint32_t total = (uint32_t)__shfl((int)inc, 63, 64) - ((uint32_t)__shfl((int)inc, 0, 64) - (uint32_t)__shfl((int)touched, 0, 64));
    if ((threadIdx.x & 63) == 0 && total > (uint32_t)FRG_BWD_HEAVY_SLOTS) heavy_waves[1 + atomicAdd(&heavy_waves[0], 1u)] = (uint32_t)(first_gaussian >> 6);
}

// Persistent workgroups of 1024 threads walk chunks of 1024 Gaussians (chunk c is
// always handled by workgroup c % gridDim.x -- the scatter kernel relies on the same
// map).  Per-tile instance counts are accumulated in an LDS histogram private to the
// workgroup and flushed ONCE, as a row of the (workgroup x tile) count matrix: no
// global atomics (the first version issued R = 16.4 M of them at C3 and ran 8x over
// its bandwidth bound).  LDS_BINS=false: T too large for LDS -> global atomics.
#define PRE_SUB 16        // Gaussians per SH transpose step
#define PRE_ROW_F4 13     // 12 float4 + 1 pad: odd stride, conflict-free ds_read_b128

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// SH colour AND d(colour)/d(direction) of the wave's 64 Gaussians from the float4-streamed coefficient rows
// (M == 16).  The wave's 64 x 48 coefficients are one contiguous stream of 768 float4: read coalesced, 16 Gaussians at
// a time through the wave's LDS buffer (software pipeline: the next needed sub-batch is in flight while the current
// one is consumed; sub-batches without a visible Gaussian are skipped).  A sub-batch is consumed by
// lane = (Gaussian g of 16, channel ch): 48 of the 64 lanes work, each sums its channel's 16 coefficients in the
// reference's left-to-right order (forward.cu:28-70: colours bit-identical) and, on the way, the three direction
// derivatives of its channel (ShDir::feed) -- the per-Gaussian backward needs those, and with them in sh_dir it never
// reads the 192-byte rows again.  (One Gaussian per lane used 16 of 64 lanes per sub-batch.)
//   dirs[k * dir_stride]  a float4 per Gaussian k of the wave in LDS, written here: unit direction + visible flag
//   sh_dir_out            global, this wave's first Gaussian: [64][9]
// Returns this lane's own colour record (only meaningful for a visible Gaussian).
// 16-byte load with the non-temporal hint (a stream that is read once)
__device__ __forceinline__ float4 load_stream(const float4* p)
{
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// float e of a sub-batch's [16][9] derivative block -> float index in the wave's row buffer, skipping every 13th float4
// (the rows' pads, which hold the colour sums at that time)
__device__ __forceinline__ int sh_dd_slot(int e) { const int q = e >> 2; return ((q + q / 12) << 2) + (e & 3); }

// The SH pass works on SLOTS of 16 (frg_common.h: sh_slot_dense / sh_slot_of).  DENSE -- most of the wave visible -- a
// slot is a lane: the rows stream in as one contiguous block per sub-batch, sub-batches without a visible Gaussian are
// skipped.  Otherwise only the VISIBLE Gaussians get slots, in rank order: a view that sees a third of the model (occlusion
// or frustum culling) reads a third of the rows in a third of the sub-batches, each row still 192 contiguous bytes (C4:
// per-Gaussian forward 0.175 -> 0.142 ms).  dirs[r].w then names the lane behind slot r (the rank permutation is pushed
// through the LDS crossbar, the invisible lanes behind the visible ones); DENSE: the visibility flag of lane r.
// Two instances of one body: the dense one is the code the uniform scene has always run.
// WITH_DIR false (r05): the derivative rows are neither summed nor stored -- the per-Gaussian backward forms them from the
// SH rows of the Gaussians that HAVE a gradient (GeomState::sh_layout bit 1 tells it), one visible Gaussian in seven at C3.
template <bool DENSE, bool WITH_DIR>
__device__ __forceinline__ float4 sh_pass(int deg, const float4* __restrict__ src, int nvalid, bool touched, uint64_t vis, int my_slot,
                                          float4* shbuf, const float4* dirs, int dir_stride, float* __restrict__ sh_dir_out)
{
    const int lane = threadIdx.x & 63;
    const int nvis = __popcll(vis);
    uint32_t need = 0;                  // sub-batches with work
    if (DENSE) {
#pragma unroll
        for (int h = 0; h < 64 / PRE_SUB; h++)
            if (vis & (0xFFFFull << (h * PRE_SUB))) need |= 1u << h;
    } else need = (1u << ((nvis + PRE_SUB - 1) / PRE_SUB)) - 1u;
    float4 pre[PRE_SUB * 12 / 64];
    auto issue = [&](int h) {
#pragma unroll
        for (int k = 0; k < PRE_SUB * 12 / 64; k++) {
            const int f = k * 64 + lane, gl = f / 12, j = f - gl * 12, r = h * PRE_SUB + gl;
            // (read once per view, 576 MB at C3: non-temporal, so that the stream does not push the records and pairs out of the caches)
            if (DENSE) pre[k] = r < nvalid ? load_stream(src + (size_t)h * PRE_SUB * 12 + f) : make_float4(0.f, 0.f, 0.f, 0.f);
            else pre[k] = r < nvis ? load_stream(src + (uint32_t)(__float_as_int(dirs[r * dir_stride].w) * 12 + j)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int h = __builtin_ctz(need);
    issue(h);
    float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
    const int g = lane >> 2, ch = lane & 3;
    const float* shw = reinterpret_cast<const float*>(shbuf) + g * (PRE_ROW_F4 * 4) + ch;   // coefficient i of (g, ch) at shw[3 i]
#pragma unroll 1
    while (h < 64 / PRE_SUB) {
#pragma unroll
        for (int k = 0; k < PRE_SUB * 12 / 64; k++) {
            const int f = k * 64 + lane, gl = f / 12, j = f - gl * 12;
            shbuf[gl * PRE_ROW_F4 + j] = pre[k];
        }
        const uint32_t rest = need >> (h + 1);
        const int hn = rest ? h + 1 + __builtin_ctz(rest) : 64 / PRE_SUB;
        if (hn < 64 / PRE_SUB) issue(hn);
        wave_sync_lds();
        const int r = h * PRE_SUB + g;                   // slot of this lane's Gaussian
        float4 dv = dirs[r * dir_stride];
        bool work = ch < 3;
        if (DENSE) work = work && dv.w != 0.0f;
        else { work = work && r < nvis; if (work) dv = dirs[__float_as_int(dv.w) * dir_stride]; }
        if (work) {
            float w[16];
            const int ncoef = sh_weights(deg, dv.x, dv.y, dv.z, w);
            const ShDir sd(deg, dv.x, dv.y, dv.z);
            float acc = 0.f, ddx = 0.f, ddy = 0.f, ddz = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (i < ncoef) {
                    const float sv = shw[3 * i];
                    if (i == 0) acc = w[0] * sv;
                    else if (i == 1 || i == 3) acc = acc - w[i] * sv;
                    else acc = acc + w[i] * sv;
                    if (WITH_DIR) sd.feed(i, sv, ddx, ddy, ddz);
                }
            }
            reinterpret_cast<float*>(shbuf + g * PRE_ROW_F4 + 12)[ch] = acc;    // the row's pad float4: raw colour sums
            // The sub-batch's derivative rows ([16][9] floats, contiguous in sh_dir) are assembled in the LDS rows just
            // consumed -- in the float4 slots that are not a pad -- and leave as 36 float4: stored straight from here
            // they were 144 scattered 4-byte requests per sub-batch.
            if (WITH_DIR) {
                float* rowf = reinterpret_cast<float*>(shbuf);
                const int e = g * 9 + ch;
                rowf[sh_dd_slot(e)] = ddx; rowf[sh_dd_slot(e + 3)] = ddy; rowf[sh_dd_slot(e + 6)] = ddz;
            }
        }
        wave_sync_lds();
        // sh_dir holds the rows BY SLOT inside the wave's block of 64 (the per-Gaussian backward derives the slot from the
        // same visibility ballot); empty slots carry whatever the LDS held: never read
        if (WITH_DIR && lane < PRE_SUB * 9 / 4) {     // (read again only by the per-Gaussian backward, a millisecond and gigabytes later)
            typedef float nt_f4 __attribute__((ext_vector_type(4)));
            const float4 v = shbuf[lane + lane / 12];
            __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4*>(sh_dir_out + (size_t)h * PRE_SUB * 9) + lane);
        }
        if (touched && (my_slot / PRE_SUB) == h) {
            const float4 c = shbuf[(my_slot % PRE_SUB) * PRE_ROW_F4 + 12];
            ShAccum sa;
            sa.acc[0] = c.x; sa.acc[1] = c.y; sa.acc[2] = c.z;
            mine = sa.finish();
        }
        wave_sync_lds();
        h = hn;
    }
    return mine;
}

// SPARSE: the kernel also carries the instance for waves of which less than three quarters are visible (launched when
// the view is expected to see a part of the model; GeomState::sh_layout tells the backward).
template <bool SPARSE>
__device__ __forceinline__ float4 sh_stream_wave(int deg, const float4* __restrict__ src, int nvalid, bool touched, float3 dir,
                                                 float4* shbuf, float4* dirs, int dir_stride, float* __restrict__ sh_dir_out, bool with_dir)
{
    const int lane = threadIdx.x & 63;
    const uint64_t vis = __ballot(touched);
    if (vis == 0ull) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int nvis = __popcll(vis);
    if (!SPARSE || sh_slot_dense(nvis)) {          // wave-uniform
        dirs[lane * dir_stride] = make_float4(dir.x, dir.y, dir.z, touched ? 1.0f : 0.0f);
        return with_dir ? sh_pass<true, true>(deg, src, nvalid, touched, vis, lane, shbuf, dirs, dir_stride, sh_dir_out)
                        : sh_pass<true, false>(deg, src, nvalid, touched, vis, lane, shbuf, dirs, dir_stride, sh_dir_out);
    }
    const int rank = __popcll(vis & ((1ull << lane) - 1ull));
    const int behind = __builtin_amdgcn_ds_permute((touched ? rank : nvis + (lane - rank)) << 2, lane);
    dirs[lane * dir_stride] = make_float4(dir.x, dir.y, dir.z, __int_as_float(behind));
    wave_sync_lds();
    return with_dir ? sh_pass<false, true>(deg, src, nvalid, touched, vis, rank, shbuf, dirs, dir_stride, sh_dir_out)
                    : sh_pass<false, false>(deg, src, nvalid, touched, vis, rank, shbuf, dirs, dir_stride, sh_dir_out);
}

// SHMODE: how the SH colour of a visible Gaussian is produced.
//   SH_INLINE  in this kernel, one strided read per coefficient (any layout)
//   SH_STREAM  in this kernel, coefficients streamed as float4 and transposed through LDS (M == 16)
//   SH_DEFER   not here: sh_color_kernel computes it on a side stream while the binning stages (scan, scatter,
//              sort -- LDS / latency bound, HBM nearly idle) run on the caller's; the blend waits for both
//   SH_STREAM_SPARSE  SH_STREAM with the instance for sparsely visible waves (sh_stream_wave<true>)
enum { SH_INLINE = 0, SH_STREAM = 1, SH_DEFER = 2, SH_STREAM_SPARSE = 3 };
// BINMODE: how the per-tile instance counts are formed.
//   BIN_WALK   every (Gaussian, tile) instance bumps its tile's bin (walk over the wave's concatenated rectangles)
//   BIN_TIGHT  the same walk, instances that provably touch no pixel of their tile dropped
//   BIN_CELLS  (cell-ordered scatter) every tile of the rectangle counts, so the counts are the 2-D prefix sum of a
//              DIFFERENCE array with four entries per Gaussian (+1 top-left, -1 right of the top-right, -1 below the
//              bottom-left, +1 diagonal): 4 LDS atomics per Gaussian, no walk; the prefix sum is linear, so it is
//              taken once, over the sum of all workgroups' arrays (scan_tiles).  Plus one bin per record cell.
enum { BIN_WALK = 0, BIN_TIGHT = 1, BIN_CELLS = 2 };
template <bool LDS_BINS, int SHMODE, int BINMODE>
__global__ void __launch_bounds__(FRG_BIN_THREADS)
preprocess_fwd_kernel(int P, ViewParams vp, const float* __restrict__ viewmatrix,
                      const float* __restrict__ projmatrix, const float* __restrict__ cam_pos,
                      const float* __restrict__ means3D, const float* __restrict__ scales,
                      const float* __restrict__ rotations, const float* __restrict__ opacities,
                      const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
                      const float* __restrict__ colors_precomp, const unsigned char* __restrict__ keep_mask, RawInputs raw,
                      int* __restrict__ radii, float4* __restrict__ xydr, float4* __restrict__ conic_opacity,
                      float4* __restrict__ rgb_clamped, uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ depth_rect,
                      uint32_t* __restrict__ bin_matrix, uint32_t* __restrict__ tile_count,
                      uint32_t* __restrict__ block_sums, Counters* __restrict__ counters, int prefiltered,
                      uint32_t* __restrict__ row_matrix, int band_w, int nbands, float* __restrict__ sh_dir,
                      uint32_t* __restrict__ heavy_waves, uint32_t* __restrict__ sh_layout)
{
    constexpr bool SH16 = SHMODE == SH_STREAM || SHMODE == SH_STREAM_SPARSE;
    constexpr bool TIGHT = BINMODE == BIN_TIGHT, CELLS = BINMODE == BIN_CELLS;
    static_assert(!CELLS || LDS_BINS, "the cell-ordered scatter needs the LDS bins");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_bins[];
    __shared__ float4 sh_lds[SH16 ? (FRG_BIN_THREADS / 64) * PRE_SUB * PRE_ROW_F4 : 1];
    __shared__ uint32_t emit_start[(FRG_BIN_THREADS / 64) * 68];
    __shared__ int4 emit_info[FRG_BIN_THREADS];
    // the chunk's 48-byte records are assembled in LDS (centre / conic first, colour after the SH sum) and leave as
    // one contiguous, fully written block per wave: piecewise 16-byte stores at a 48-byte stride cost 10 % of the kernel
    __shared__ float4 rec_lds[FRG_BIN_THREADS * FRG_REC];
    static_assert(sizeof(sh_lds) + sizeof(emit_start) + sizeof(emit_info) + sizeof(rec_lds) <= FRG_BIN_STATIC_LDS,
                  "static LDS of preprocess_fwd_kernel grew: lower FRG_BIN_MAX_LDS_TILES");
    static_assert(FRG_BIN_STATIC_LDS + 4 * FRG_BIN_MAX_LDS_TILES + 256 <= 160 * 1024, "LDS bins + static arrays exceed the CU's 160 KiB");
    const int T = vp.gx * vp.gy;
    ViewMats vmx;
    load_view_mats(viewmatrix, projmatrix, cam_pos, vmx);
    const int nchunks = (P + FRG_BIN_THREADS - 1) / FRG_BIN_THREADS;
    // bins [T, T + ncells): visible Gaussians by the cell (tile row, band of band_w tile columns) of their rectangle's
    // first tile (reorder_kernel's counts)
    const int ncells = CELLS ? vp.gy * nbands : 0;
    const int nbins = T + ncells;
    if (LDS_BINS)
        for (int t = threadIdx.x; t < nbins; t += FRG_BIN_THREADS) lds_bins[t] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        heavy_waves[0] = 0;   // (filled where point_offsets is finished)
        // bit 0: sh_dir rows by rank in sparsely visible waves; bit 1: no sh_dir rows at all (only the float4-streamed pass has that form)
        if (SHMODE != SH_DEFER) *sh_layout = (SHMODE == SH_STREAM_SPARSE ? 1u : 0u) | ((vp.sh_no_dir && SHMODE != SH_INLINE) ? 2u : 0u);    // (SH_DEFER: sh_color_kernel says)
    }
    // the chunk totals of this workgroup's chunks are accumulated with atomics below
    for (int c = blockIdx.x + (int)threadIdx.x * (int)gridDim.x; c < nchunks; c += FRG_BIN_THREADS * (int)gridDim.x) block_sums[c] = 0;
    __threadfence_block();
    __syncthreads();
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int idx = c * FRG_BIN_THREADS + threadIdx.x;
        uint32_t touched = 0;
        int rx0 = 0, ry0 = 0, rw = 1;
        float3 dir = make_float3(0.f, 0.f, 1.f);
        float4* rec = rec_lds + threadIdx.x * FRG_REC;
        rec[0] = rec[1] = rec[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < P) {
            int radius_i, x0, y0, x1, y1;
            float depth = 0.f;
            touched = preprocess_one(idx, vp, vmx, means3D, scales, rotations, opacities, shs, cov3D_precomp,
                                     colors_precomp, keep_mask, raw, rec, counters, prefiltered,
                                     radius_i, x0, y0, x1, y1, dir, depth);
            radii[idx] = radius_i;
            tiles_touched[idx] = touched;
            rx0 = x0; ry0 = y0; rw = x1 - x0;
            if (touched) {
                if (CELLS) {
                    atomicAdd(&lds_bins[T + y0 * nbands + x0 / band_w], 1u);
                    atomicAdd(&lds_bins[y0 * vp.gx + x0], 1u);         // unsigned wrap-around is harmless
                    if (x1 < vp.gx) atomicAdd(&lds_bins[y0 * vp.gx + x1], 0xFFFFFFFFu);
                    if (y1 < vp.gy) {
                        atomicAdd(&lds_bins[y1 * vp.gx + x0], 0xFFFFFFFFu);
                        if (x1 < vp.gx) atomicAdd(&lds_bins[y1 * vp.gx + x1], 1u);
                    }
                }
                // three planes of P words (a lane's three stores each join its neighbours' in one line)
                depth_rect[idx] = __float_as_uint(depth);
                depth_rect[(size_t)P + idx] = (uint32_t)x0 | ((uint32_t)y0 << 16);
                depth_rect[2 * (size_t)P + idx] = (uint32_t)x1 | ((uint32_t)y1 << 16);
            }
        }
        if (!CELLS) {   // per-tile instance counts
            const int wave = threadIdx.x >> 6;
            wave_for_each_instance(touched, rx0, ry0, rw, 0u, 0u, emit_start + wave * 68, emit_info + wave * 64, vp.gx,
                                   [&](int owner, int t, int tx, int ty, uint32_t, uint32_t) {
                                       if (TIGHT) {   // centre and conic of the owner: its record, still in LDS
                                           const float4* orec = rec_lds + (wave * 64 + owner) * FRG_REC;
                                           const float4 c2 = orec[0];
                                           if (!tile_hit(c2.x, c2.y, orec[1], tx, ty)) return;
                                       }
                                       if (LDS_BINS) atomicAdd(&lds_bins[t], 1u);   // ds_add_u32
                                       else atomicAdd(&tile_count[t], 1u);
                                   });
        }
        // ---- colour ----
        if (colors_precomp) {
            if (touched)
                rec[2] = make_float4(colors_precomp[3 * idx], colors_precomp[3 * idx + 1], colors_precomp[3 * idx + 2],
                                               __uint_as_float(0u));
        } else if (SHMODE != SH_DEFER) {
            if (SH16) {
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
                const int idx0 = c * FRG_BIN_THREADS + wave * 64;
                // the direction of every Gaussian of the wave waits in the (still empty) colour slot of its record
                const float4 col = sh_stream_wave<SHMODE == SH_STREAM_SPARSE>(vp.D, reinterpret_cast<const float4*>(shs) + (size_t)idx0 * 12, min(64, P - idx0),
                                                  touched != 0, dir, sh_lds + wave * (PRE_SUB * PRE_ROW_F4),
                                                  rec_lds + (wave * 64) * FRG_REC + 2, FRG_REC, sh_dir + (size_t)idx0 * 9, !vp.sh_no_dir);
                (void)lane;
                rec[2] = touched ? col : make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (touched) {
                float w[16];
                const int ncoef = sh_weights(vp.D, dir.x, dir.y, dir.z, w);
                const ShDir sd(vp.D, dir.x, dir.y, dir.z);
                ShAccum sa;
                sa.acc[0] = sa.acc[1] = sa.acc[2] = 0.f;
                float dd[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};   // [x|y|z][channel]
                const float* sh = shs + (size_t)idx * vp.M * 3;
#pragma unroll
                for (int e = 0; e < 48; e++) {
                    const int i = e / 3, ch = e % 3;
                    if (i < ncoef) { const float sv = sh[e]; sa.add(i, ch, w[i], sv); sd.feed(i, sv, dd[0][ch], dd[1][ch], dd[2][ch]); }
                }
                rec[2] = sa.finish();
#pragma unroll
                for (int k = 0; k < 9; k++) sh_dir[(size_t)idx * 9 + k] = dd[k / 3][k % 3];
            }
        }
        {   // the wave's 64 records leave as 3 x 64 consecutive float4
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const size_t first = (size_t)(c * FRG_BIN_THREADS + wave * 64) * FRG_REC;
            const size_t limit = (size_t)P * FRG_REC;
            wave_sync_lds();
#pragma unroll
            for (int j = 0; j < FRG_REC; j++)
                if (first + j * 64 + lane < limit) xydr[first + j * 64 + lane] = rec_lds[wave * 64 * FRG_REC + j * 64 + lane];
            wave_sync_lds();
        }
        // chunk total: one atomic per wave into this workgroup's own (pre-zeroed) entry.  No
        // workgroup barrier in the chunk loop: the 16 waves drift apart, so one wave's SH
        // streaming overlaps another's covariance math instead of all waves changing phase together.
        const uint32_t wtotal = wave_incl_scan(touched, threadIdx.x & 63);
        if ((threadIdx.x & 63) == 63 && wtotal) atomicAdd(&block_sums[c], wtotal);
    }
    if (LDS_BINS) {
        __syncthreads();
        uint32_t* row = bin_matrix + (size_t)blockIdx.x * T;
        for (int t = threadIdx.x; t < T; t += FRG_BIN_THREADS) row[t] = lds_bins[t];
        if (CELLS)
            for (int r = threadIdx.x; r < ncells; r += FRG_BIN_THREADS) row_matrix[(size_t)blockIdx.x * ncells + r] = lds_bins[T + r];
    }
}

// SH colour of the visible Gaussians as a kernel of its own (SH_DEFER): same arithmetic, same summation
// order, same float4 stream through wave-private LDS as the in-kernel form -- the colours are bit-identical.
// One wave = 64 consecutive Gaussians; nothing here depends on the binning, so the kernel runs on a side
// stream beside scan / scatter / sort and only the blend joins it.
#define SHC_THREADS 256
template <bool SH16, bool SPARSE>
__global__ void __launch_bounds__(SHC_THREADS)
sh_color_kernel(int P, int D, int M, const float* __restrict__ cam_pos, const float* __restrict__ means3D,
                const int* __restrict__ radii, const float* __restrict__ shs, float4* __restrict__ rgb_clamped,
                float* __restrict__ sh_dir, uint32_t* __restrict__ sh_layout, int sh_no_dir)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *sh_layout = ((SH16 && SPARSE) ? 1u : 0u) | ((SH16 && sh_no_dir) ? 2u : 0u);
    __shared__ float4 sh_lds[SH16 ? (SHC_THREADS / 64) * PRE_SUB * PRE_ROW_F4 : 1];
    __shared__ float4 dir_lds[SH16 ? SHC_THREADS : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx0 = blockIdx.x * SHC_THREADS + wave * 64;
    const int idx = idx0 + lane;
    const bool touched = idx < P && radii[idx] > 0;
    if (__ballot(touched) == 0ull) return;                      // wave-uniform: nothing visible here
    float3 dir = make_float3(0.f, 0.f, 1.f);
    if (touched) {   // unit view direction (forward.cu:25-27), the expressions of preprocess_one
        const float cx = cam_pos[0], cy = cam_pos[1], cz = cam_pos[2];
        const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        float dx = p.x - cx, dy = p.y - cy, dz = p.z - cz;
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        dir = make_float3(dx / len, dy / len, dz / len);
    }
    if (SH16) {
        const float4 col = sh_stream_wave<SPARSE>(D, reinterpret_cast<const float4*>(shs) + (size_t)idx0 * 12, min(64, P - idx0), touched, dir,
                                          sh_lds + wave * (PRE_SUB * PRE_ROW_F4), dir_lds + wave * 64, 1, sh_dir + (size_t)idx0 * 9, !sh_no_dir);
        if (touched) rgb_clamped[FRG_REC * idx] = col;
    } else if (touched) {
        float w[16];
        const int ncoef = sh_weights(D, dir.x, dir.y, dir.z, w);
        const ShDir sd(D, dir.x, dir.y, dir.z);
        ShAccum sa;
        sa.acc[0] = sa.acc[1] = sa.acc[2] = 0.f;
        float dd[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        const float* sh = shs + (size_t)idx * M * 3;
#pragma unroll
        for (int e = 0; e < 48; e++) {
            const int i = e / 3, ch = e % 3;
            if (i < ncoef) { const float sv = sh[e]; sa.add(i, ch, w[i], sv); sd.feed(i, sv, dd[0][ch], dd[1][ch], dd[2][ch]); }
        }
        rgb_clamped[FRG_REC * idx] = sa.finish();
#pragma unroll
        for (int k = 0; k < 9; k++) sh_dir[(size_t)idx * 9 + k] = dd[k / 3][k % 3];
    }
}

// The two single-workgroup scans of the binning stage.  Both are latency-bound: every thread owns SCAN_K CONSECUTIVE
// elements, loads them in one batch of independent requests, and the workgroup runs one scan over the thread sums.
//   scan_chunks  exclusive scan of the per-chunk sums (in place), total -> counters.num_rendered, overflow flag
//   scan_tiles   per-tile totals -> ranges [start,end), empty tiles (0,0) exactly as the reference's memset +
//                identifyTileRanges leave them (rasterizer_impl.cu:310-317); max -> counters.max_tile_count; the sort's
//                work lists; with use_segs == 1 the per-segment sums become per-segment start offsets for colbase_kernel
// scan_kernel runs both; with cell-ordered records they ride as extra workgroups of colsum_kernel and reorder_kernel
// (the tile scan then runs beside the reorder instead of in front of it).
struct ScanShared {
    uint32_t ovf, carry, maxc;
    uint32_t wtot[16];
    uint32_t cls[FRG_SORT_CLASSES];
    uint32_t sub[FRG_SORT_CLASSES * 8], cur[FRG_SORT_CLASSES * 8];
    uint32_t nempty;
};
#define SCAN_K 10

template <int NT>
__device__ __forceinline__ void scan_chunks(ScanShared& sh, int nchunks, uint32_t* __restrict__ block_sums,
                                            Counters* __restrict__ counters, uint32_t capacity, Mailbox* mail, uint32_t seq)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { sh.carry = 0; sh.ovf = 0; }
    __syncthreads();
    for (int base = 0; base < nchunks; base += NT * SCAN_K) {
        const int per = min(SCAN_K, (min(nchunks - base, NT * SCAN_K) + NT - 1) / NT);   // elements per thread in this round
        const int lo = base + tid * per;
        uint32_t v[SCAN_K], sum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_K; k++) {
            v[k] = (k < per && lo + k < nchunks) ? block_sums[lo + k] : 0u;
            sum += v[k];
        }
        const uint32_t inc = wave_incl_scan(sum, lane);
        if (lane == 63) sh.wtot[wave] = inc;
        __syncthreads();
        uint32_t woff = 0, all = 0;
        for (int w = 0; w < NT / 64; w++) { const uint32_t x = sh.wtot[w]; if (w < wave) woff += x; all += x; }
        const uint32_t carry = sh.carry;
        uint32_t run = carry + woff + inc - sum;
#pragma unroll
        for (int k = 0; k < SCAN_K; k++)
            if (k < per && lo + k < nchunks) { block_sums[lo + k] = run; run += v[k]; }
        __syncthreads();
        if (tid == 0) sh.carry = carry + all;
        __syncthreads();
    }
    if (tid == 0) {
        counters->num_rendered = sh.carry;
        // more instances than the caller's binning buffer holds (deferred-counters forward):
        // every tile list is left empty, the frame renders as background and is redone
        // (written either way: in the LDS-bins paths nothing clears the counters before the forward)
        const uint32_t ovf = (capacity && sh.carry > capacity) ? 1u : 0u;
        sh.ovf = ovf; counters->overflow = ovf;
        counters->carved_R = capacity ? capacity : sh.carry;    // what the host carves the binning chunk with (api.hip)
        if (mail) { mail->num_rendered = sh.carry; mailbox_post(&mail->seq_r, seq); }
    }
    __syncthreads();
}

// The tile scan is ONE workgroup executing its code once: the time goes to instruction fetch (straight-line unrolled
// code ran at an instruction-cache miss per 64 bytes, ~30 us at C3, twice that beside a streaming kernel), not to
// data.  So only the load batch is unrolled; the totals then sit in LDS (tot, T words) and everything else is a
// rolled loop of a few dozen instructions.  USE_SEGS: 0 totals in tile_count | 1 sum the segments and turn them into
// start offsets for colbase_kernel | 2 sum the segments only | 3 the summed difference arrays are in tile_count (one word
// per tile from colsum_kernel: the load batch of this latency chain is an eighth of form 2's).
template <int NT, int USE_SEGS>
__device__ __forceinline__ void scan_tiles(ScanShared& sh, uint32_t* tot, bool ovf, int T, int gx, uint32_t* __restrict__ tile_count,
                                           uint32_t* __restrict__ seg_sums, uint2* __restrict__ ranges,
                                           uint32_t* __restrict__ class_tiles, Counters* __restrict__ counters, uint32_t tight,
                                           Mailbox* mail, uint32_t seq)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { sh.carry = 0; sh.maxc = 0; sh.nempty = 0; }
    if (tid < FRG_SORT_CLASSES) sh.cls[tid] = 0;
    if (tid < FRG_SORT_CLASSES * 8) { sh.sub[tid] = 0; sh.cur[tid] = 0; }
    // totals -> LDS: coalesced, all requests of a round in flight together.  (Images of more tiles than the LDS holds
    // have their totals in tile_count already -- USE_SEGS == 0 -- and are scanned from there: tot == tile_count.)
    if (tot == tile_count) {
        if (ovf) for (int i = tid; i < T; i += NT) tile_count[i] = 0;
    } else
    for (int base = 0; base < T; base += NT * SCAN_K) {
        uint32_t v[SCAN_K];
#pragma unroll
        for (int k = 0; k < SCAN_K; k++) {
            const int i = base + k * NT + tid;
            v[k] = 0;
            if (i < T) {
                if (USE_SEGS == 1 || USE_SEGS == 2) {
#pragma unroll
                    for (int sg = 0; sg < FRG_BIN_SEGS; sg++) v[k] += seg_sums[(size_t)sg * T + i];
                } else v[k] = tile_count[i];
            }
        }
#pragma unroll
        for (int k = 0; k < SCAN_K; k++) {
            const int i = base + k * NT + tid;
            if (i < T) tot[i] = ovf ? 0u : v[k];
        }
    }
    __syncthreads();
    if (USE_SEGS >= 2) {
        // the staged totals are the summed DIFFERENCE arrays of the preprocess (BIN_CELLS): instances per tile = their
        // 2-D prefix sum, along x (one wave per tile row) and then along y (one thread per tile column)
        const int gy = T / gx;
        for (int row = wave; row < gy; row += NT / 64) {
            uint32_t carry = 0;
            for (int xb = 0; xb < gx; xb += 64) {
                const int x = xb + lane;
                const uint32_t v = x < gx ? tot[row * gx + x] : 0u;
                const uint32_t inc = wave_incl_scan(v, lane) + carry;
                if (x < gx) tot[row * gx + x] = inc;
                carry = (uint32_t)__shfl((int)inc, 63, 64);
            }
        }
        __syncthreads();
        for (int x = tid; x < gx; x += NT) {
            uint32_t run = tot[x];
#pragma unroll 1
            for (int row = 1; row < gy; row++) { run += tot[row * gx + x]; tot[row * gx + x] = run; }
        }
        __syncthreads();
    }
    // Wave w owns the tiles [w * span, (w + 1) * span), span a multiple of 64, LANE = TILE: the totals are read and
    // ranges / tile_count stored 64 consecutive tiles at a time, and a tile's start is one DPP scan away (every thread
    // owning ~7 consecutive tiles walked two dependent per-thread loops: scan stage 0.061 -> 0.057 ms at C3).
    constexpr int NW = NT / 64;
    const int span = ((T + NW - 1) / NW + 63) / 64 * 64;
    const int w_lo = min(T, wave * span), w_hi = min(T, w_lo + span);
    uint32_t sum = 0, local_max = 0;
#pragma unroll 1
    for (int t = w_lo + lane; t < w_hi; t += 64) { const uint32_t c = tot[t]; sum += c; local_max = max(local_max, c); }
    const uint32_t wsum = wave_incl_scan_dpp(sum);
    if (lane == 63) sh.wtot[wave] = wsum;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local_max = max(local_max, (uint32_t)__shfl_xor((int)local_max, d, 64));
    if (lane == 0) atomicMax(&sh.maxc, local_max);
    __syncthreads();
    uint32_t carry = 0;
    for (int w = 0; w < wave; w++) carry += sh.wtot[w];
#pragma unroll 1
    for (int t0 = w_lo; t0 < w_hi; t0 += 64) {          // wave-uniform trip count
        const int i = t0 + lane;
        const bool mine = i < w_hi;
        const uint32_t c = mine ? tot[i] : 0u;
        const uint32_t inc = wave_incl_scan_dpp(c);
        const uint32_t excl = carry + inc - c;
        carry += (uint32_t)__shfl((int)inc, 63, 64);
        if (!mine) continue;
        if (USE_SEGS || ovf) tile_count[i] = c;
        ranges[i] = c ? make_uint2(excl, excl + c) : make_uint2(0u, 0u);
        if (c) atomicAdd(&sh.sub[sort_subclass_of(c)], 1u);
        if (USE_SEGS == 1) {  // segment s of tile i starts at excl + sum of earlier segments (colbase_kernel)
            uint32_t r2 = excl;
#pragma unroll 1
            for (int sg = 0; sg < FRG_BIN_SEGS; sg++) {
                const uint32_t c2 = seg_sums[(size_t)sg * T + i];
                seg_sums[(size_t)sg * T + i] = r2;
                r2 += c2;
            }
        }
    }
    __syncthreads();
    // work lists of the sort: only non-empty tiles, grouped by size class and, inside a class,
    // by size in eight descending buckets (longest tiles are dispatched first: the last round of
    // workgroups then holds the short ones instead of a straggler)
    if (tid < FRG_SORT_CLASSES) {
        uint32_t r3 = 0;
        for (int k = 0; k < 8; k++) { const uint32_t c = sh.sub[tid * 8 + k]; sh.sub[tid * 8 + k] = r3; r3 += c; }
        sh.cls[tid] = r3;
    }
    __syncthreads();
#pragma unroll 1
    for (int i = tid; i < T; i += NT) {
        const uint32_t c = tot[i];
        if (!c) {   // the empty tiles, listed behind the classes: the forward blend can then walk the tiles longest list first
            class_tiles[(size_t)FRG_SORT_CLASSES * T + atomicAdd(&sh.nempty, 1u)] = (uint32_t)i;
            continue;
        }
        const int key = sort_subclass_of(c);
        class_tiles[(size_t)(key >> 3) * T + sh.sub[key] + atomicAdd(&sh.cur[key], 1u)] = (uint32_t)i;
    }
    if (tid == 0) { counters->max_tile_count = sh.maxc; counters->tight_binning = tight; }
    if (tid < FRG_SORT_CLASSES) counters->class_count[tid] = sh.cls[tid];
    if (mail && tid == 0) {
        // (num_rendered, filtered, overflow: written by earlier kernels or by this workgroup before a barrier)
        Counters c;
        c.num_rendered = __hip_atomic_load(&counters->num_rendered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.filtered = __hip_atomic_load(&counters->filtered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.overflow = __hip_atomic_load(&counters->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.max_tile_count = sh.maxc;
        c.tight_binning = tight;
        c.num_visible = 0;
        c.carved_R = __hip_atomic_load(&counters->carved_R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.bwd_seg_log = 0; c.fwd_flags = 0;     // (the forward blend's stamps: not made yet, and nothing the host reads here)
        for (int k = 0; k < FRG_SORT_CLASSES; k++) c.class_count[k] = sh.cls[k];
        mail->c = c;
        mailbox_post(&mail->seq_c, seq);
    }
}

__global__ void __launch_bounds__(1024)
scan_kernel(int nchunks, uint32_t* __restrict__ block_sums, int T, uint32_t* __restrict__ tile_count,
            uint32_t* __restrict__ seg_sums, int use_segs, uint2* __restrict__ ranges, uint32_t* __restrict__ class_tiles,
            Counters* __restrict__ counters, uint32_t capacity, uint32_t tight, int lds_tot, Mailbox* mail, uint32_t seq)
{
    __shared__ ScanShared sh;
    extern __shared__ __attribute__((aligned(16))) uint32_t scan_tot[];    // T words (lds_tot)
    scan_chunks<1024>(sh, nchunks, block_sums, counters, capacity, mail, seq);
    if (use_segs) scan_tiles<1024, 1>(sh, scan_tot, sh.ovf != 0, T, 0, tile_count, seg_sums, ranges, class_tiles, counters, tight, mail, seq);
    else scan_tiles<1024, 0>(sh, lds_tot ? scan_tot : tile_count, sh.ovf != 0, T, 0, tile_count, seg_sums, ranges, class_tiles, counters, tight, mail, seq);
}

// Column sums of the count matrix.
//   one_total == 0 (scatter in the caller's order): split into FRG_BIN_SEGS row segments, seg_sums[s][t] = sum over the
//                  rows of segment s; grid (ceil(T / 256), FRG_BIN_SEGS);
//   one_total == 1 (cell-ordered records): ONE total per tile -> tile_count[t] (the summed difference arrays; the tile scan
//                  turns them into counts): a workgroup takes 64 tiles, its four waves a quarter of the rows each; grid
//                  (ceil(T / 64), 3), and two more rows of workgroups ride along:
//                    blockIdx.y == 1  one wave per cell turns the counts of visible Gaussians per (workgroup, cell) into
//                                     each workgroup's offset inside the cell's block of records (the workgroups of one
//                                     XCD next to each other) and leaves the cell's total in row_total (reorder_kernel
//                                     scans the totals);
//                    blockIdx.y == 2  one workgroup scans the chunk totals (scan_chunks).
__global__ void __launch_bounds__(256)
colsum_kernel(int T, int nrows, const uint32_t* __restrict__ bin_matrix, uint32_t* __restrict__ seg_sums,
              uint32_t* __restrict__ row_matrix, uint32_t* __restrict__ row_total, int gy,
              int nchunks, uint32_t* __restrict__ block_sums, Counters* __restrict__ counters, uint32_t capacity,
              Mailbox* mail, uint32_t seq, uint32_t* __restrict__ tile_fill, uint32_t* __restrict__ tile_work,
              uint32_t* __restrict__ bwd_cnt, int one_total, uint32_t* __restrict__ tile_count)
{
    if (one_total && blockIdx.y == 2) {
        if (blockIdx.x != 0) return;
        __shared__ ScanShared sh;
        scan_chunks<256>(sh, nchunks, block_sums, counters, capacity, mail, seq);
        return;
    }
    if (one_total && blockIdx.y == 1) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int r = blockIdx.x * 4 + wave;
        if (r >= gy) return;
        // four workgroups per lane (nrows <= FRG_BIN_MAX_BLOCKS = 256), XCD-major when nrows is a multiple of 8
        const int per = (nrows % FRG_NUM_XCD == 0) ? nrows / FRG_NUM_XCD : 0;
        uint32_t c[4], s4 = 0;
        int w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = 4 * lane + i;
            w[i] = per ? (k / per) + FRG_NUM_XCD * (k % per) : k;
            c[i] = k < nrows ? row_matrix[(size_t)w[i] * gy + r] : 0u;
            s4 += c[i];
        }
        const uint32_t inc = wave_incl_scan(s4, lane);
        uint32_t run = inc - s4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (4 * lane + i < nrows) row_matrix[(size_t)w[i] * gy + r] = run;
            run += c[i];
        }
        if (lane == 63) row_total[r] = inc;
        return;
    }
    // the counters of the backward blend's item lists (filled by the forward blend's tile workgroups) start at zero
    if (blockIdx.y == 0 && blockIdx.x == 0)
        for (int i = threadIdx.x; i < FRG_NUM_XCD * (FRG_BWD_LEN_BUCKETS + 1); i += 256) bwd_cnt[i] = 0u;
    if (one_total) {
        __shared__ uint32_t part[4][64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int t = blockIdx.x * 64 + lane;
        uint32_t s = 0;
        if (t < T) {
#pragma unroll 8
            for (int r = wave; r < nrows; r += 4) s += bin_matrix[(size_t)r * T + t];
        }
        part[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && t < T) {
            tile_count[t] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
            // the scatter's fill cursor and the forward blend's depth mark of every tile start at zero: cleared here, on
            // the way, instead of by a memset launch in front of every forward
            tile_fill[t] = 0u; tile_work[t] = 0u;
        }
        return;
    }
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    if (blockIdx.y == 0) { tile_fill[t] = 0u; tile_work[t] = 0u; }
    // segment s = the workgroups the dispatcher places on XCD s (round robin, FRG_BIN_SEGS == 8):
    // a tile's runs written by one XCD are then adjacent in memory, so partially written lines
    // are completed inside that XCD's L2 instead of being written back piecemeal by eight L2s
    uint32_t s = 0;
#pragma unroll 8
    for (int r = blockIdx.y; r < nrows; r += FRG_BIN_SEGS) s += bin_matrix[(size_t)r * T + t];
    seg_sums[(size_t)blockIdx.y * T + t] = s;
}

// Turns the count matrix into the base matrix in place: base[b][t] = first position
// in point_list/pairs that workgroup b will write for tile t.
__global__ void __launch_bounds__(256)
colbase_kernel(int T, int nrows, uint32_t* __restrict__ bin_matrix, const uint32_t* __restrict__ seg_start)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    uint32_t run = seg_start[(size_t)blockIdx.y * T + t];
#pragma unroll 8
    for (int r = blockIdx.y; r < nrows; r += FRG_BIN_SEGS) {
        const uint32_t c = bin_matrix[(size_t)r * T + t];
        bin_matrix[(size_t)r * T + t] = run;
        run += c;
    }
}

// Same chunk -> workgroup map as preprocess.  Finishes the inclusive scan
// (point_offsets, identical to the reference's cub InclusiveSum output) and scatters
// (depth bits, index) into the Gaussian's tiles at base[workgroup][tile] + LDS rank.
// The order inside a tile segment is arbitrary here; the LDS sort orders by
// (depth, index), which equals the reference's stable sort of index-ordered keys
// (rasterizer_impl.cu:98-108, :303-308).
template <bool LDS_BINS, bool TIGHT>
__global__ void __launch_bounds__(FRG_BIN_THREADS)
scatter_kernel(int P, int gx, int gy, const uint32_t* __restrict__ depth_rect, const float4* __restrict__ xydr,
               const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ chunk_prefix,
               uint32_t* __restrict__ point_offsets, const uint32_t* __restrict__ bin_matrix,
               const uint2* __restrict__ ranges, uint32_t* __restrict__ tile_fill, uint2* __restrict__ pairs,
               const Counters* __restrict__ counters, const float4* __restrict__ conic_opacity, uint32_t* __restrict__ heavy_waves)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_bins[];
    // wave-uniform: the binning buffer is too small for this frame.  Nothing is scattered (every tile list
    // was left empty by scan_kernel), but point_offsets is still finished: a backward issued before the
    // caller has seen the overflow walks it, and must find this frame's scan, not a stale one
    const bool overflow = counters->overflow != 0;
    __shared__ uint32_t wsum[FRG_BIN_THREADS / 64];
    __shared__ uint32_t emit_start[(FRG_BIN_THREADS / 64) * 68];
    __shared__ int4 emit_info[FRG_BIN_THREADS];
    __shared__ float2 emit_xy[TIGHT ? FRG_BIN_THREADS : 1];
    __shared__ float4 emit_co[TIGHT ? FRG_BIN_THREADS : 1];
    const int T = gx * gy;
    if (LDS_BINS && !overflow) {
        const uint32_t* row = bin_matrix + (size_t)blockIdx.x * T;
        for (int t = threadIdx.x; t < T; t += FRG_BIN_THREADS) lds_bins[t] = row[t];
        __syncthreads();
    }
    const int nchunks = (P + FRG_BIN_THREADS - 1) / FRG_BIN_THREADS;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int idx = c * FRG_BIN_THREADS + threadIdx.x;
        const uint32_t touched = idx < P ? tiles_touched[idx] : 0u;
        uint32_t total;
        const uint32_t inc = block_incl_scan<FRG_BIN_THREADS / 64>(touched, wsum, &total);
        if (idx < P) point_offsets[idx] = chunk_prefix[c] + inc;
        note_heavy_wave(inc, touched, c * FRG_BIN_THREADS + (int)(threadIdx.x & ~63u), heavy_waves);
        if (overflow) continue;
        int x0 = 0, y0 = 0, x1 = 1, y1 = 0;
        uint32_t dbits = 0;
        if (touched) {
            dbits = depth_rect[idx];
            const uint32_t lo = depth_rect[(size_t)P + idx], hi = depth_rect[2 * (size_t)P + idx];
            x0 = (int)(lo & 0xFFFFu); y0 = (int)(lo >> 16); x1 = (int)(hi & 0xFFFFu); y1 = (int)(hi >> 16);
            (void)y1;   // (the walk needs the rectangle's origin and width only)
            if (TIGHT) { const float4 g = xydr[FRG_REC * idx]; emit_xy[threadIdx.x] = make_float2(g.x, g.y); emit_co[threadIdx.x] = conic_opacity[FRG_REC * idx]; }
        }
        const int wave = threadIdx.x >> 6;
        const uint32_t idx0 = (uint32_t)(c * FRG_BIN_THREADS + wave * 64);
        wave_for_each_instance(touched, x0, y0, x1 - x0, dbits, 0u, emit_start + wave * 68, emit_info + wave * 64, gx,
                               [&](int owner, int t, int tx, int ty, uint32_t depth_bits, uint32_t) {
                                   if (TIGHT) {
                                       const float2 c2 = emit_xy[wave * 64 + owner];
                                       if (!tile_hit(c2.x, c2.y, emit_co[wave * 64 + owner], tx, ty)) return;
                                   }
                                   uint32_t pos;
                                   if (LDS_BINS) pos = atomicAdd(&lds_bins[t], 1u);             // ds_add_rtn_u32
                                   else pos = ranges[t].x + atomicAdd(&tile_fill[t], 1u);
                                   pairs[pos] = make_uint2(depth_bits, idx0 + (uint32_t)owner);
                               });
    }
}

// ---- scatter over cell-ordered records ---------------------------------------------------------
// In the caller's order a chunk of 1024 Gaussians hits ~6500 different tiles all over the image: every 128-byte line
// of the pairs array is written 16 times over the whole duration of the scatter, no cache can hold 131 MB of open
// lines, and they reach HBM as 32-byte sectors (3.8x write amplification, the stage ran at 15 % of its roofline).
// With the Gaussians of the C3 scene handed over sorted by screen row the very same kernel took 0.085 instead of
// 0.222 ms -- so the library orders its own 16-byte scatter records first, by the CELL (tile row, band of a few tile
// columns) of the rectangle's first tile:
//   reorder_kernel       same chunk -> workgroup map as preprocess (whose per-workgroup cell counts, scanned by the
//                        extra workgroups of colsum_kernel, are its write positions: no atomics on global memory);
//                        also finishes point_offsets; one extra workgroup scans the tile totals beside it
//   scatter_rows_kernel  every workgroup takes one contiguous share of the records (a few cells: some dozens of
//                        tiles): counts its instances per tile in LDS, reserves one run per touched tile with a single
//                        atomic on the tile's fill cursor, emits.  The shares of one XCD are consecutive (dealt like
//                        the tile bands of the blend), so the lines of a tile segment are completed inside one L2.
// The order of the pairs inside a tile segment is arbitrary (and, with the run reservations, not the same from run to
// run); the sort orders them by (depth, index), a total order: point_list is deterministic.
__global__ void __launch_bounds__(FRG_BIN_THREADS, 8)   // 8 waves per SIMD = two workgroups per CU: the scan workgroup must fit beside a reorder workgroup
reorder_kernel(int P, int nblocks, int ncells, int band_w, int nbands, const uint32_t* __restrict__ depth_rect, const uint32_t* __restrict__ tiles_touched,
               const uint32_t* __restrict__ chunk_prefix, uint32_t* __restrict__ point_offsets,
               const uint32_t* __restrict__ row_matrix, const uint32_t* __restrict__ row_total, uint4* __restrict__ row_records,
               Counters* __restrict__ counters,
               int T, int gx, uint32_t* __restrict__ tile_count, uint32_t* __restrict__ seg_sums, uint2* __restrict__ ranges,
               uint32_t* __restrict__ class_tiles, uint32_t tight, uint32_t* __restrict__ heavy_waves, Mailbox* mail, uint32_t seq)
{
    const int my_block = (int)blockIdx.x;     // this workgroup's row of the count matrices
    if (my_block == nblocks) {
        // one more workgroup than the reorder needs: the scan of the tile totals (ranges, the sort's work lists,
        // the counters the host reads back) -- a single workgroup's latency chain, beside the reorder instead of in
        // front of it.  Nothing in the reorder depends on it.
        __shared__ ScanShared sh;
        extern __shared__ __attribute__((aligned(16))) uint32_t scan_tot[];    // T words (dynamic LDS of this launch)
        scan_tiles<FRG_BIN_THREADS, 3>(sh, scan_tot, counters->overflow != 0, T, gx, tile_count, seg_sums, ranges, class_tiles, counters, tight, mail, seq);
        return;
    }
    __shared__ uint32_t cursor[FRG_MAX_TILE_ROWS];
    __shared__ uint32_t wsum[FRG_BIN_THREADS / 64];
    {   // first record of every cell = exclusive scan of the cell totals (at most FRG_MAX_TILE_ROWS = the workgroup
        // size; every workgroup repeats these few hundred additions rather than wait for another launch)
        const uint32_t v = (int)threadIdx.x < ncells ? row_total[threadIdx.x] : 0u;
        uint32_t all;
        const uint32_t inc = block_incl_scan<FRG_BIN_THREADS / 64>(v, wsum, &all);
        if ((int)threadIdx.x < ncells) cursor[threadIdx.x] = inc - v + row_matrix[(size_t)my_block * ncells + threadIdx.x];
        if (my_block == 0 && threadIdx.x == 0) counters->num_visible = all;
        __syncthreads();
    }
    const int nchunks = (P + FRG_BIN_THREADS - 1) / FRG_BIN_THREADS;
    for (int c = my_block; c < nchunks; c += nblocks) {
        const int idx = c * FRG_BIN_THREADS + threadIdx.x;
        const uint32_t touched = idx < P ? tiles_touched[idx] : 0u;
        uint3 dr = make_uint3(0u, 0u, 0u);
        if (touched) dr = make_uint3(depth_rect[idx], depth_rect[(size_t)P + idx], depth_rect[2 * (size_t)P + idx]);
        uint32_t total;
        const uint32_t inc = block_incl_scan<FRG_BIN_THREADS / 64>(touched, wsum, &total);
        if (idx < P) point_offsets[idx] = chunk_prefix[c] + inc;
        note_heavy_wave(inc, touched, c * FRG_BIN_THREADS + (int)(threadIdx.x & ~63u), heavy_waves);
        if (touched) {
            const uint32_t pos = atomicAdd(&cursor[(dr.y >> 16) * nbands + (dr.y & 0xFFFFu) / band_w], 1u);     // ds_add_rtn_u32
            row_records[pos] = make_uint4(dr.x, (uint32_t)idx, dr.y, dr.z);
        }
    }
}

// Records of one workgroup = FRG_ROWS_SUB sub-slices of 1024 at most (launch_scatter sizes the grid accordingly).
#define FRG_ROWS_SUB 8
__global__ void __launch_bounds__(FRG_BIN_THREADS)
scatter_rows_kernel(int T, int gx, int gy, const uint4* __restrict__ row_records, const uint2* __restrict__ ranges,
                    uint32_t* __restrict__ tile_fill, uint2* __restrict__ pairs, const Counters* __restrict__ counters, int ablate,
                    const uint32_t* __restrict__ heavy_waves, Mailbox* mail, uint32_t seq)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_bins[];
    __shared__ uint32_t emit_start[(FRG_BIN_THREADS / 64) * 68];
    __shared__ int4 emit_info[FRG_BIN_THREADS];
    __shared__ int row_lo, row_hi;
    // (the list of heavy waves was finished by reorder_kernel)
    if (mail && blockIdx.x == 0 && threadIdx.x == 0) {
        mail->visible = counters->num_visible;
        __threadfence_system();
        __hip_atomic_store(&mail->heavy_post, ((unsigned long long)seq << 32) | heavy_waves[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the binning buffer is too small for this frame (deferred-counters forward): every tile list was left empty
    if (counters->overflow != 0) return;
    // Every workgroup takes one contiguous share of the records, a multiple of 1024 (a few cells: some dozens of
    // tiles), and reserves ONE run per touched tile.  Workgroup b runs on XCD b % 8: XCD x takes the shares
    // [x * per_xcd, (x + 1) * per_xcd), so that a tile's pairs are written through one L2.
    // (The grid is sized on the host, which does not know the number of records: the shares are cut HERE, and dealt to the
    // XCDs evenly whatever the grid -- with gridDim / 8 shares per XCD a grid a quarter too large left two XCDs idle:
    // 0.093 instead of 0.083 ms at C3.)
    const uint32_t nvis = counters->num_visible;
    const uint32_t share = ((nvis + gridDim.x - 1) / gridDim.x + FRG_BIN_THREADS - 1) / FRG_BIN_THREADS * FRG_BIN_THREADS;
    const uint32_t nshares = share ? (nvis + share - 1) / share : 0u;
    const uint32_t per_xcd = (nshares + FRG_NUM_XCD - 1) / FRG_NUM_XCD;
    if (blockIdx.x / FRG_NUM_XCD >= per_xcd) return;        // workgroup-uniform: the grid has more workgroups than there are shares
    const uint32_t first = ((blockIdx.x % FRG_NUM_XCD) * per_xcd + blockIdx.x / FRG_NUM_XCD) * share;
    if (first >= nvis) return;                       // workgroup-uniform
    const uint32_t last = min(nvis, first + share);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = threadIdx.x; t < T; t += FRG_BIN_THREADS) lds_bins[t] = 0;
    if (threadIdx.x == 0) { row_lo = gy; row_hi = 0; }
    __syncthreads();
    struct Rec { int x0, y0, x1, y1; uint32_t depth, index, touched; bool valid; };
    auto load = [&](uint32_t base) -> Rec {
        Rec r;
        const uint32_t i = base + threadIdx.x;
        r.valid = i < last;
        const uint4 rec = r.valid ? row_records[i] : make_uint4(0u, 0u, 0u, 0u);
        r.x0 = (int)(rec.z & 0xFFFFu); r.y0 = (int)(rec.z >> 16); r.x1 = (int)(rec.w & 0xFFFFu); r.y1 = (int)(rec.w >> 16);
        r.depth = rec.x; r.index = rec.y;
        r.touched = r.valid ? (uint32_t)((r.x1 - r.x0) * (r.y1 - r.y0)) : 0u;
        return r;
    };
    // ---- counts per tile ----
    // Every tile of a rectangle counts: the per-tile counts are the 2-D prefix sum of a difference array with four
    // entries per Gaussian (+1 top-left, -1 right of the top-right, -1 below the bottom-left, +1 diagonal; entries
    // beyond the end of a tile row or below the last row are not needed) -- 4 LDS atomics per Gaussian and two short
    // scans instead of a walk over its 6.5 instances.  Unsigned wrap-around is harmless.
    int lo = gy, hi = 0;
    for (uint32_t base = first; base < last; base += FRG_BIN_THREADS) {
        const Rec r = load(base);
        if (!r.valid) continue;
        lo = min(lo, r.y0); hi = max(hi, r.y1);
        atomicAdd(&lds_bins[r.y0 * gx + r.x0], 1u);
        if (r.x1 < gx) atomicAdd(&lds_bins[r.y0 * gx + r.x1], 0xFFFFFFFFu);
        if (r.y1 < gy) {
            atomicAdd(&lds_bins[r.y1 * gx + r.x0], 0xFFFFFFFFu);
            if (r.x1 < gx) atomicAdd(&lds_bins[r.y1 * gx + r.x1], 1u);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo = min(lo, __shfl_xor(lo, d, 64)); hi = max(hi, __shfl_xor(hi, d, 64)); }
    if (lane == 0) { atomicMin(&row_lo, lo); atomicMax(&row_hi, hi); }
    __syncthreads();
    const int rlo = row_lo, rhi = max(row_hi, row_lo);   // tile rows this workgroup's records touch
    for (int row = rlo + wave; row < rhi; row += FRG_BIN_THREADS / 64) {      // along x: one wave per row
        uint32_t carry = 0;
        for (int xb = 0; xb < gx; xb += 64) {
            const int x = xb + lane;
            const uint32_t v = x < gx ? lds_bins[row * gx + x] : 0u;
            const uint32_t inc = wave_incl_scan(v, lane) + carry;
            if (x < gx) lds_bins[row * gx + x] = inc;
            carry = (uint32_t)__shfl((int)inc, 63, 64);
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < gx; x += FRG_BIN_THREADS) {                   // along y: one thread per column
        uint32_t run = lds_bins[rlo * gx + x];
        for (int row = rlo + 1; row < rhi; row++) { run += lds_bins[row * gx + x]; lds_bins[row * gx + x] = run; }
    }
    __syncthreads();
    // ---- one run per touched tile: the bin becomes the position of the run's first pair ----
    for (int t = rlo * gx + (int)threadIdx.x; t < rhi * gx; t += FRG_BIN_THREADS) {
        const uint32_t c = lds_bins[t];
        if (c) lds_bins[t] = ranges[t].x + atomicAdd(&tile_fill[t], c);
    }
    __syncthreads();
    // ---- emit ----
    for (uint32_t base = first; base < last; base += FRG_BIN_THREADS) {
        const Rec r = load(base);                    // (second read of the records: they come from the L2)
        wave_for_each_instance(r.touched, r.x0, r.y0, r.x1 - r.x0, r.depth, r.index, emit_start + wave * 68, emit_info + wave * 64, gx,
                               [&](int, int t, int, int, uint32_t depth_bits, uint32_t index) {
                                   const uint32_t pos = atomicAdd(&lds_bins[t], 1u);            // ds_add_rtn_u32
                                   if (!(ablate & 4) || pos == 0xFFFFFFFFu) pairs[pos] = make_uint2(depth_bits, index);   // (ablate: timing experiments)
                               });
    }
}

// rasterizer_impl.cu:54-66
__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ viewmatrix,
                    unsigned char* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float vm[16];
#pragma unroll
    for (int i = 0; i < 16; i++) vm[i] = viewmatrix[i];
    const float3 pv = xform43(p, vm);
    present[idx] = (pv.z <= 0.2f) ? 0 : 1;
}

// ---- host launchers -----------------------------------------------------------
// workgroups of the cell-ordered scatter (tuning knob: frg_set_option("rows_grid")), 2 per CU by default
int g_rows_grid = 0;     // 0: by the model's size (launch_scatter); > 0: timing experiments

// The scatter runs over cell-ordered records (reorder_kernel + scatter_rows_kernel) in the reference-identical
// binning mode; tight binning keeps the scatter in the caller's order (it would evaluate the per-instance tile test in
// both passes of the new kernel and gather the centre / conic of every record: measured slower).
static bool cell_order(const ImageState& img, const ViewParams& vp) { return img.row_order && !vp.tight; }

static int bin_blocks(int P)
{
    const int nchunks = (P + FRG_BIN_THREADS - 1) / FRG_BIN_THREADS;
    return nchunks < FRG_BIN_MAX_BLOCKS ? nchunks : FRG_BIN_MAX_BLOCKS;
}

template <typename K>
static hipError_t allow_big_lds(K kernel, size_t bytes)
{
    if (bytes <= 16 * 1024) return hipSuccess;   // (static LDS of these kernels is large: ask early)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <bool LDS_BINS, int SHMODE, int BINMODE>
static hipError_t launch_pre_variant(int P, const ViewParams& vp, const FwdInputs& in, int* radii, const GeomState& g,
                                     const ImageState& img, int prefiltered, hipStream_t s)
{
    const int T = vp.gx * vp.gy;
    const int nb = bin_blocks(P);
    const size_t lds = LDS_BINS ? (size_t)(T + (BINMODE == BIN_CELLS ? img.ncells : 0)) * 4 : 0;
    hipError_t e = allow_big_lds(preprocess_fwd_kernel<LDS_BINS, SHMODE, BINMODE>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((preprocess_fwd_kernel<LDS_BINS, SHMODE, BINMODE>), dim3(nb), dim3(FRG_BIN_THREADS), lds, s, P, vp, in.viewmatrix,
                       in.projmatrix, in.cam_pos, in.means3D, in.scales, in.rotations, in.opacities, in.shs,
                       in.cov3D_precomp, in.colors_precomp, in.keep_mask, in.raw, radii, g.xydr, g.conic_opacity, g.rgb_clamped,
                       g.tiles_touched, g.depth_rect, img.bin_matrix, img.tile_count, g.block_sums, img.counters, prefiltered,
                       BINMODE == BIN_CELLS ? img.row_matrix : nullptr, img.band_w, img.nbands, g.sh_dir, g.heavy_waves, g.sh_layout);
    return hipGetLastError();
}

// float4-streamed SH needs the reference's usual layout: 16 coefficients per channel, 16-byte aligned
static bool sh_streamable(const FwdInputs& in, const ViewParams& vp)
{
    return in.shs && vp.M == 16 && (reinterpret_cast<uintptr_t>(in.shs) % 16 == 0);
}

hipError_t launch_preprocess_fwd(int P, const ViewParams& vp, const FwdInputs& in, int* radii, const GeomState& g,
                                 const ImageState& img, int prefiltered, bool defer_sh, hipStream_t s)
{
    const int mode = (defer_sh && in.shs) ? SH_DEFER : sh_streamable(in, vp) ? SH_STREAM : SH_INLINE;
#define FRG_PRE(L, S) (vp.tight ? launch_pre_variant<L, S, BIN_TIGHT>(P, vp, in, radii, g, img, prefiltered, s) \
                                : launch_pre_variant<L, S, BIN_WALK>(P, vp, in, radii, g, img, prefiltered, s))
#define FRG_PRE_CELLS(S) launch_pre_variant<true, S, BIN_CELLS>(P, vp, in, radii, g, img, prefiltered, s)
    // (the instance for sparsely visible views only with the cell-ordered scatter, the default path)
    if (cell_order(img, vp)) return mode == SH_DEFER ? FRG_PRE_CELLS(SH_DEFER) : mode == SH_STREAM ? (vp.sparse_sh ? FRG_PRE_CELLS(SH_STREAM_SPARSE) : FRG_PRE_CELLS(SH_STREAM)) : FRG_PRE_CELLS(SH_INLINE);
    if (img.lds_bins) return mode == SH_DEFER ? FRG_PRE(true, SH_DEFER) : mode == SH_STREAM ? FRG_PRE(true, SH_STREAM) : FRG_PRE(true, SH_INLINE);
    return mode == SH_DEFER ? FRG_PRE(false, SH_DEFER) : mode == SH_STREAM ? FRG_PRE(false, SH_STREAM) : FRG_PRE(false, SH_INLINE);
#undef FRG_PRE
#undef FRG_PRE_CELLS
}

hipError_t launch_sh_color(int P, const ViewParams& vp, const FwdInputs& in, const int* radii, const GeomState& g, hipStream_t s)
{
    if (!in.shs || P <= 0) return hipSuccess;
    const dim3 grid((P + SHC_THREADS - 1) / SHC_THREADS), block(SHC_THREADS);
    if (sh_streamable(in, vp)) {
        if (vp.sparse_sh)
            hipLaunchKernelGGL((sh_color_kernel<true, true>), grid, block, 0, s, P, vp.D, vp.M, in.cam_pos, in.means3D, radii, in.shs, g.rgb_clamped, g.sh_dir, g.sh_layout, vp.sh_no_dir);
        else
            hipLaunchKernelGGL((sh_color_kernel<true, false>), grid, block, 0, s, P, vp.D, vp.M, in.cam_pos, in.means3D, radii, in.shs, g.rgb_clamped, g.sh_dir, g.sh_layout, vp.sh_no_dir);
    } else
        hipLaunchKernelGGL((sh_color_kernel<false, false>), grid, block, 0, s, P, vp.D, vp.M, in.cam_pos, in.means3D, radii, in.shs, g.rgb_clamped, g.sh_dir, g.sh_layout, vp.sh_no_dir);
    return hipGetLastError();
}

hipError_t launch_scan(int P, const ViewParams& vp, const GeomState& g, const ImageState& img, uint32_t capacity, hipStream_t s,
                       Mailbox* mail, uint32_t seq)
{
    const int T = vp.gx * vp.gy;
    const int nchunks = (P + FRG_BIN_THREADS - 1) / FRG_BIN_THREADS;
    const int nb = bin_blocks(P);
    const bool cells = cell_order(img, vp);
    if (img.lds_bins)
        hipLaunchKernelGGL(colsum_kernel, cells ? dim3(std::max((T + 63) / 64, (img.ncells + 3) / 4), 3) : dim3((T + 255) / 256, FRG_BIN_SEGS),
                           dim3(256), 0, s, T, nb, img.bin_matrix, img.seg_sums, img.row_matrix, img.row_start, img.ncells,
                           nchunks, g.block_sums, img.counters, capacity, mail, seq, img.tile_fill, img.tile_work, img.bwd_cnt,
                           cells ? 1 : 0, img.tile_count);
    if (cells) {
        // the records in cell order (+ point_offsets); its extra workgroup scans the tile totals
        hipLaunchKernelGGL(reorder_kernel, dim3(nb + 1), dim3(FRG_BIN_THREADS), (size_t)T * 4, s, P, nb, img.ncells, img.band_w, img.nbands,
                           g.depth_rect, g.tiles_touched, g.block_sums, g.point_offsets, img.row_matrix, img.row_start, g.row_records,
                           img.counters, T, vp.gx, img.tile_count, img.seg_sums, img.ranges, img.class_tiles, (uint32_t)vp.tight, g.heavy_waves, mail, seq);
        return hipGetLastError();
    }
    // the tile totals sit in LDS (T words) when they fit
    const bool lds_tot = (size_t)T * 4 <= 128 * 1024;
    if (lds_tot) {
        hipError_t e = allow_big_lds(scan_kernel, (size_t)T * 4);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), lds_tot ? (size_t)T * 4 : 0, s, nchunks, g.block_sums, T, img.tile_count, img.seg_sums,
                       img.lds_bins ? 1 : 0, img.ranges, img.class_tiles, img.counters, capacity, (uint32_t)vp.tight, lds_tot ? 1 : 0, mail, seq);
    // per-workgroup scatter bases of the scatter in the caller's order
    if (img.lds_bins)
        hipLaunchKernelGGL(colbase_kernel, dim3((T + 255) / 256, FRG_BIN_SEGS), dim3(256), 0, s, T, nb, img.bin_matrix, img.seg_sums);
    return hipGetLastError();
}

hipError_t launch_scatter(int P, const ViewParams& vp, const int* radii, const GeomState& g, const ImageState& img,
                          const BinningState& b, hipStream_t s, int ablate, Mailbox* mail, uint32_t seq)
{
    const int T = vp.gx * vp.gy;
    const int nb = bin_blocks(P);
    if (cell_order(img, vp)) {   // (launch_scan has reordered the records)
        const size_t lds = (size_t)T * 4;
        // two workgroups per CU, each with one contiguous share of the records (their number is only known on the
        // device: at most P); more workgroups only when a share would exceed FRG_ROWS_SUB sub-slices
        const int need = (P + FRG_ROWS_SUB * FRG_BIN_THREADS - 1) / (FRG_ROWS_SUB * FRG_BIN_THREADS);
        // ... and fewer for small models: a share's fixed costs (its LDS difference array over the tiles, one reserved run
        // per touched tile) do not shrink with it -- C2 (100 k Gaussians, 70 k records), same box: 512 workgroups 19 us,
        // 256 16, 128 15, 64 18
        // ... and more than two per CU for the large ones (r05, shares dealt evenly to the XCDs whatever the grid; same box, scatter
        // stage at C3 / C4): 512 workgroups 0.088 / 0.098 ms, 640 0.081 / 0.098, 768 0.080 / 0.089, 1024 0.081 / 0.089, 1536 0.078 / 0.089
        const int by_size = std::min(1024, std::max(64, P / 768));
        const int grid = ((std::max(need, g_rows_grid > 0 ? g_rows_grid : by_size) + FRG_NUM_XCD - 1) / FRG_NUM_XCD) * FRG_NUM_XCD;
        hipError_t e = allow_big_lds(scatter_rows_kernel, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(FRG_BIN_THREADS), lds, s, T, vp.gx, vp.gy, g.row_records,
                           img.ranges, img.tile_fill, b.pairs, img.counters, ablate, g.heavy_waves, mail, seq);
        return hipGetLastError();
    }
#define FRG_SCATTER(L, TI, LDS)                                                                                          \
    hipLaunchKernelGGL((scatter_kernel<L, TI>), dim3(nb), dim3(FRG_BIN_THREADS), LDS, s, P, vp.gx, vp.gy, g.depth_rect, g.xydr,    \
                       g.tiles_touched, g.block_sums, g.point_offsets, img.bin_matrix, img.ranges, img.tile_fill, b.pairs,  \
                       img.counters, g.conic_opacity, g.heavy_waves)
    if (img.lds_bins) {
        const size_t lds = (size_t)T * 4;
        hipError_t e = vp.tight ? allow_big_lds(scatter_kernel<true, true>, lds) : allow_big_lds(scatter_kernel<true, false>, lds);
        if (e != hipSuccess) return e;
        if (vp.tight) FRG_SCATTER(true, true, lds); else FRG_SCATTER(true, false, lds);
    } else {
        if (vp.tight) FRG_SCATTER(false, true, 0); else FRG_SCATTER(false, false, 0);
    }
#undef FRG_SCATTER
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, unsigned char* present, hipStream_t s)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
    return hipGetLastError();
}

}  // namespace frg
