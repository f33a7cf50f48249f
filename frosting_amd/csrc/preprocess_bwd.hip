// Per-Gaussian backward: deterministic reduction of the blend partials, then the
// reference's K8 + K9 fused into one streaming kernel:
//   computeCov2DCUDA   backward.cu:144-274   (conic -> cov2D -> cov3D and mean)
//   preprocessCUDA bwd backward.cu:346-396   (projection path)
//   computeColorFromSH backward.cu:20-139    (SH and view-direction path)
//   computeCov3D bwd   backward.cu:278-341   (scale / quaternion)
// Every gradient row is written exactly once (zeros for culled Gaussians), so the
// caller does not pre-zero ~300 B/Gaussian as the reference must
// (rasterize_points.cu:151-159).
//
// wave64 structure (each wave owns 64 consecutive Gaussians, no workgroup barrier):
//   1. slot reduction: the wave's Gaussians own one contiguous run of Gaussian-major
//      slots; the 64 lanes walk that run 64 slots at a time (coalesced), find each
//      slot's owner by binary search over the lane offsets in LDS, drop slots whose
//      tile never processed them (tile cutoff key), and sum per owner with a
//      segmented shuffle scan -- fixed order, no atomics, no per-lane loop whose
//      length is the largest Gaussian of the wave;
//   2. SH coefficients (192 B/Gaussian) are read, and their gradients written, as
//      wave-contiguous float4 streams transposed through LDS 16 Gaussians at a time
//      instead of 48 stride-192 dword accesses per lane.
#include "gauss_math.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace frg {


#define BWD_THREADS 256
#define BWD_SUB 16                       // Gaussians per SH transpose step
#define BWD_ROW_F4 13                    // 12 float4 of SH + 1 pad (odd stride: conflict-free b128)
#define BWD_LDS_WORDS (68 + 256 + 576 + BWD_SUB * BWD_ROW_F4 * 4 + 64)
// Slots are reduced in WINDOWS of BWD_WIN slots of the wave's run; a wave whose 64 Gaussians own more than
// four windows (a handful of near-camera Gaussians covering hundreds of tiles each: one wave walked 38 000 slots while
// the average wave has 400, and the kernel waited for it -- 0.99 instead of 0.28 ms on the clustered scene) leaves its
// Gaussians to a second launch, beside this one, in which a whole 16-wave workgroup takes the windows 16 at a time.
// Per-window sums are added in window order in both forms: the same bits either way.
#define BWD_WIN 896u                     // 10 bits of position, 6 bits of owner
static_assert(FRG_BWD_HEAVY_SLOTS == 4 * BWD_WIN, "hand-over threshold: four windows");
#define BWD_HEAVY_WAVES 16

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool SH16, bool HEAVY>
__global__ void __launch_bounds__(HEAVY ? BWD_HEAVY_WAVES * 64 : BWD_THREADS, 4)
preprocess_bwd_kernel(int P, ViewParams vp, const float* __restrict__ viewmatrix,
                      const float* __restrict__ projmatrix, const float* __restrict__ cam_pos,
                      const float* __restrict__ means3D, const int* __restrict__ radii,
                      const float* __restrict__ shs, const float* __restrict__ scales,
                      const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
                      const float4* __restrict__ xydr, const float4* __restrict__ rgb_clamped,
                      const float4* __restrict__ conic_opacity,
                      const uint32_t* __restrict__ point_offsets, const uint2* __restrict__ cutoff,
                      const Counters* __restrict__ counters, const float* __restrict__ slots,
                      float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
                      float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
                      float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot, int ablate,
                      RawInputs raw, float* __restrict__ dL_dshell_logits, float* __restrict__ dL_dshell_verts,
                      const float* __restrict__ sh_dir, int flags, const uint32_t* __restrict__ heavy,
                      const uint32_t* __restrict__ sh_layout, float* __restrict__ sums, unsigned char* __restrict__ row_live,
                      unsigned long long* __restrict__ live_masks, float* __restrict__ view_dir_terms, int first_block)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[(HEAVY ? BWD_HEAVY_WAVES : BWD_THREADS / 64) * BWD_LDS_WORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* lds = lds_all + wave * BWD_LDS_WORDS;
    uint32_t* own_start = lds;                                   // [65] slot start relative to the wave's first slot
    int4* own_info = reinterpret_cast<int4*>(lds + 68);          // [64] x0, y0, rect width, depth bits
    float* wacc = reinterpret_cast<float*>(lds + 68 + 256);      // [64][9] per-owner sums of the current window
    float4* shbuf = reinterpret_cast<float4*>(lds + 68 + 256 + 576);  // [BWD_SUB][BWD_ROW_F4]
    uint32_t* own_base = lds + 68 + 256 + 576 + BWD_SUB * BWD_ROW_F4 * 4;   // [64] first slot of the Gaussian, relative to the wave's first

    // heavy[0] = number of listed waves, heavy[1 + k] = their wave numbers (GeomState::heavy_waves, left by the forward)
  uint32_t heavy_item = HEAVY ? blockIdx.x : 0u;
  if (HEAVY && heavy_item >= heavy[0]) return;                  // (usually: nothing was handed over)
  do {   // HEAVY: the workgroup strides over the handed-over waves; otherwise once
    if (HEAVY && heavy_item != blockIdx.x) __syncthreads();     // the previous item's LDS contents are dead
    const int idx0 = HEAVY ? (int)heavy[1 + heavy_item] * 64 : ((int)blockIdx.x + first_block) * BWD_THREADS + wave * 64;       // first Gaussian of this wave
    const int idx = idx0 + lane;
    const bool valid = idx < P;
    ViewMats vmx;
    load_view_mats(viewmatrix, projmatrix, cam_pos, vmx);

    const int radius = valid ? radii[idx] : 0;
    const bool visible = radius > 0;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t clamp_bits = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (visible) {
        g = xydr[FRG_REC * idx];
        clamp_bits = __float_as_uint(rgb_clamped[FRG_REC * idx].w);
        tile_rect(g.x, g.y, radius, vp.gx, vp.gy, x0, y0, x1, y1);
    }
    const int sh_row = idx0 + sh_slot_of(__ballot(visible), lane, (*sh_layout & 1u) != 0u);     // (frg_common.h: by lane or by rank among the wave's visible Gaussians)
    // ---- 1. slot reduction ------------------------------------------------------
    const uint32_t incl = valid ? point_offsets[idx] : 0u;
    const uint32_t base = valid ? (idx == 0 ? 0u : point_offsets[idx - 1]) : 0u;
    const uint32_t wave_base = (uint32_t)__shfl((int)base, 0, 64);
    uint32_t S_all = valid ? incl - wave_base : 0u;  // the wave's slots: run length = max over valid lanes
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) S_all = max(S_all, (uint32_t)__shfl_xor((int)S_all, d, 64));
    // The run that is REDUCED holds the slots of the Gaussians the backward blend marked (FRG_REACHED_MASK: it staged one of
    // their instances with a non-empty quadrant mask) -- a third of the slots at C3; the others' slots hold zeros or
    // nothing.  Position sl of that run belongs to owner o = the last lane with own_start[o] <= sl and is the slot
    // own_base[o] + (sl - own_start[o]) of the wave's slots.
    const uint32_t n_own = (visible && (clamp_bits & FRG_REACHED_MASK)) ? incl - base : 0u;
    uint32_t S = n_own;                              // inclusive scan over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)S, d, 64);
        if (lane >= d) S += up;
    }
    own_start[lane] = S - n_own;
    own_base[lane] = base - wave_base;
    S = (uint32_t)__shfl((int)S, 63, 64);            // (invalid lanes: n_own = 0, their start is the run's end)
    if (lane == 0) own_start[64] = S;
    own_info[lane] = make_int4(x0, y0, x1 - x0, (int)__float_as_uint(g.z));
    // Slots of instances that provably touch no pixel of their tile (tile_hit false) hold nothing: with tight binning
    // they were never binned, otherwise the blend backward found their quadrant mask empty and wrote zeros.  They are
    // skipped in BOTH modes -- half of the processed slots, and the list of slots that ARE read is then the same
    // list in both modes, so the order of the additions, hence every gradient bit, does not depend on the mode.
    // The owner's centre and conic sit in the (still unused) SH transpose buffer during this phase.
    float4* own_co = shbuf;                                         // [64]
    float2* own_xy = reinterpret_cast<float2*>(shbuf + 64);         // [64]
    own_co[lane] = visible ? conic_opacity[FRG_REC * idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    own_xy[lane] = make_float2(g.x, g.y);
    float part[FRG_SLOT_FLOATS];                                  // this lane's Gaussian: sum over the windows, in window order
#pragma unroll
    for (int c = 0; c < FRG_SLOT_FLOATS; c++) part[c] = 0.0f;
    wave_fence();
    // Only a quarter of the slots were processed by the blend backward (the tiles stop at saturation); the rest hold
    // nothing.  Two passes over windows of BWD_WIN slots of the wave's run:
    //   A  per slot: owner by binary search over the lane offsets in LDS (own_start / own_info do not change), its
    //      tile, the tile's cutoff key -> processed or not; the processed ones are compacted (ballot) into a list of
    //      16-bit entries {position in the window, owner};
    //   B  64 list entries at a time: the 36-byte rows, the segmented sum per owner, the accumulators.
    // The nine loads and the ~130-instruction segmented scan run on a quarter of the batches instead of all of them.
    uint16_t* live = reinterpret_cast<uint16_t*>(shbuf + 96);           // [BWD_WIN] behind own_co / own_xy
    if (ablate & 1) S = 0;   // TIMING EXPERIMENT ONLY (frg_set_option("ablate")): no slot reduction
    if (flags & FRG_PBW_FROM_SUMS) S = 0;   // phase 2 of a two-call backward: the sums were left by phase 1
    if (counters->fwd_flags & FRG_FWD_ONLY) S = 0;   // the forward kept nothing for a backward (and the blend backward wrote no slot): zero rows
    const uint32_t nwin = (S + BWD_WIN - 1) / BWD_WIN;                   // wave-uniform
    // on the forward's list: the 16-wave launch has it -- unless the host skipped that launch (FRG_PBW_NO_HEAVY_LAUNCH:
    // its forward posted "no such wave"), in which case a wave that does own that many slots is reduced right here,
    // window after window: whatever the host believed, no Gaussian is left without its gradients
    // (the forward listed the waves by ALL their slots: the same number decides here)
    if (!HEAVY && S_all > (uint32_t)FRG_BWD_HEAVY_SLOTS && !(flags & FRG_PBW_NO_HEAVY_LAUNCH)) return;
    // HEAVY: round r gives window 16 r + wave to this wave; non-heavy: window after window
    for (uint32_t wr = 0; wr < (HEAVY ? (nwin + BWD_HEAVY_WAVES - 1) / BWD_HEAVY_WAVES : nwin); wr++) {
        const uint32_t w0 = (HEAVY ? wr * BWD_HEAVY_WAVES + (uint32_t)wave : wr) * BWD_WIN;
#pragma unroll
        for (int c = 0; c < FRG_SLOT_FLOATS; c++) wacc[lane * FRG_SLOT_FLOATS + c] = 0.0f;
        wave_fence();
      if (w0 < S) {
        const uint32_t wend = min(S, w0 + BWD_WIN);
        uint32_t nlive = 0;                                              // wave-uniform
        // ---- A: two batches per step, so that their cutoff loads are in flight together ----
        for (uint32_t s0 = w0; s0 < wend; s0 += 128) {
            bool ok[2];
            int own[2];
            uint2 cut[2];
            uint32_t dbits[2], gid[2];
            bool live_lane[2], binned[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t sl = s0 + 64 * u + lane;
                live_lane[u] = sl < wend;
                int owner = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int mid = owner + step;
                    if (mid < 64 && own_start[mid] <= sl) owner = mid;
                }
                own[u] = owner;
                cut[u] = make_uint2(0u, 0u); dbits[u] = 0xFFFFFFFFu; gid[u] = 0; binned[u] = true;
                if (live_lane[u]) {
                    const int4 info = own_info[owner];
                    const uint32_t k = sl - own_start[owner];
                    uint32_t ry, rx;
                    rect_divmod(k, (uint32_t)info.z, ry, rx);
                    const int tx = info.x + (int)rx, ty = info.y + (int)ry;
                    cut[u] = cutoff[ty * vp.gx + tx];
                    dbits[u] = (uint32_t)info.w; gid[u] = (uint32_t)(idx0 + owner);
                    const float2 c2 = own_xy[owner];
                    binned[u] = tile_hit(c2.x, c2.y, own_co[owner], tx, ty);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                // processed by the blend backward iff (depth, index) <= the tile's cutoff key
                ok[u] = live_lane[u] && binned[u] && (dbits[u] < cut[u].x || (dbits[u] == cut[u].x && gid[u] <= cut[u].y));
                const uint64_t m = __builtin_amdgcn_ballot_w64(ok[u]);
                if (ok[u]) live[nlive + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((s0 + 64 * u + lane - w0) | ((uint32_t)own[u] << 10));
                nlive += (uint32_t)__popcll(m);
            }
        }
        wave_fence();
        // ---- B ----
        auto fetch = [&](uint32_t b0, int& owner, bool& in_run, float (&part)[FRG_SLOT_FLOATS]) {   // (part: the batch's rows, not the outer totals)
            in_run = b0 + lane < nlive;
            owner = 64 + lane;  // unique: never merges with a neighbour
#pragma unroll
            for (int c = 0; c < FRG_SLOT_FLOATS; c++) part[c] = 0.0f;
            if (in_run) {
                const uint32_t e = live[b0 + lane];
                owner = (int)(e >> 10);
                const uint32_t sl = w0 + (e & 1023u);
                const float* sp = slots + (size_t)(wave_base + own_base[owner] + (sl - own_start[owner])) * FRG_SLOT_STRIDE;
#pragma unroll
                for (int c = 0; c < FRG_SLOT_FLOATS; c++) part[c] = sp[c];
            }
        };
        // software pipeline: the next batch's rows are in flight during the scan
        int owner_n = 0;
        bool in_run_n = false;
        float part_n[FRG_SLOT_FLOATS];
        if (nlive > 0) fetch(0, owner_n, in_run_n, part_n);
        for (uint32_t b0 = 0; b0 < nlive; b0 += 64) {
            int owner = owner_n;
            const bool in_run = in_run_n;
            float part[FRG_SLOT_FLOATS];
#pragma unroll
            for (int c = 0; c < FRG_SLOT_FLOATS; c++) part[c] = part_n[c];
            if (b0 + 64 < nlive) fetch(b0 + 64, owner_n, in_run_n, part_n);
            // Segmented inclusive scan over the lanes (owners are non-decreasing with the lane), on DPP moves:
            // four row_shr steps inside the 16-lane rows, then row_bcast:15 / row_bcast:31 carry the last lane of a
            // row (of the lower half) into the lanes above that continue its owner's run.  A ds_bpermute shuffle
            // costs ~24 cycles per wave on gfx950, a DPP move ~7 (tools/micro/pk_rate.hip): 60 shuffles per batch
            // of 64 slots were the largest single item of this kernel.  A lane whose source does not exist (start
            // of the row) or belongs to another owner multiplies what it receives by 0: fma(v, 1, p) == v + p.
            float tk[6];
            {
                const int own_i = owner;
#define FRG_SEG_STEP(I, CTRL, ROWMASK)                                                                              \
                { const int o_up = __builtin_amdgcn_update_dpp(-1, own_i, CTRL, ROWMASK, 0xf, false);                   \
                  tk[I] = (o_up == own_i) ? 1.0f : 0.0f; }
                FRG_SEG_STEP(0, 0x111, 0xf) FRG_SEG_STEP(1, 0x112, 0xf) FRG_SEG_STEP(2, 0x114, 0xf) FRG_SEG_STEP(3, 0x118, 0xf)
                FRG_SEG_STEP(4, 0x142, 0xa) FRG_SEG_STEP(5, 0x143, 0xc)
#undef FRG_SEG_STEP
            }
#define FRG_SEG_ADD(I, CTRL, ROWMASK)                                                                               \
            _Pragma("unroll") for (int c = 0; c < FRG_SLOT_FLOATS; c++) {                                               \
                const float v_up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part[c]), CTRL, ROWMASK, 0xf, false)); \
                part[c] = __builtin_fmaf(v_up, tk[I], part[c]);                                                         \
            }
            FRG_SEG_ADD(0, 0x111, 0xf) FRG_SEG_ADD(1, 0x112, 0xf) FRG_SEG_ADD(2, 0x114, 0xf) FRG_SEG_ADD(3, 0x118, 0xf)
            FRG_SEG_ADD(4, 0x142, 0xa) FRG_SEG_ADD(5, 0x143, 0xc)
#undef FRG_SEG_ADD
            const int o_next = __shfl_down(owner, 1, 64);
            if (in_run && (lane == 63 || o_next != owner)) {
#pragma unroll
                for (int c = 0; c < FRG_SLOT_FLOATS; c++) wacc[owner * FRG_SLOT_FLOATS + c] += part[c];
            }
            wave_fence();
        }
      }
        // the window's sums join the totals in window order
        if (HEAVY) {
            __syncthreads();
            if (wave == 0)
                for (uint32_t j = 0; j < BWD_HEAVY_WAVES && (wr * BWD_HEAVY_WAVES + j) * BWD_WIN < S; j++) {
                    const float* wj = reinterpret_cast<const float*>(lds_all + j * BWD_LDS_WORDS + 68 + 256);
#pragma unroll
                    for (int c = 0; c < FRG_SLOT_FLOATS; c++) part[c] += wj[lane * FRG_SLOT_FLOATS + c];
                }
            __syncthreads();
        } else {
            wave_fence();
#pragma unroll
            for (int c = 0; c < FRG_SLOT_FLOATS; c++) part[c] += wacc[lane * FRG_SLOT_FLOATS + c];
            wave_fence();
        }
    }
    if (HEAVY && wave != 0) continue;      // (workgroup-level loop over the handed-over waves; wave 0 does the per-Gaussian part)
    // The backward in two calls (frg_backward_args::phase): phase 1 stops here -- the nine sums go to the workspace, and
    // dL_dcolor, complete after the reduction, is written: with shs given and dL_dsh == NULL it is the clamp-masked colour
    // gradient, the payload of the factored view-parallel exchange, which can travel while phase 2 computes.
    if (flags & FRG_PBW_SUMS_ONLY) {
        if (live_masks) {
            // The slot-sum exchange packs the rows of these Gaussians (slot_exchange.hip): has_grad of phase 2 -- and, beside
            // the sums, the three view-direction terms dd = d(colour)/d(direction) . masked dRGB that section 4 below forms
            // (the only use phase 2 makes of the forward's sh_dir / of the 192-byte SH row): with them in the packet no rank
            // has to read another view's SH rows to run this view's chain.
            bool any = false;
#pragma unroll
            for (int c = 0; c < FRG_SLOT_FLOATS; c++) any |= part[c] != 0.0f;
            any &= visible;
            const uint64_t m = __ballot(any);
            if (lane == 0) live_masks[idx0 / 64] = m;
            if (any && shs) {
                float shd[9];
                if (SH16 && (*sh_layout & 2u)) {       // the forward left no sh_dir: from the SH row, in the forward's order of additions
#pragma unroll
                    for (int k = 0; k < 9; k++) shd[k] = 0.0f;
                    const float3 mm = param_mean(means3D, raw, idx);
                    const float dox = mm.x - vmx.campos[0], doy = mm.y - vmx.campos[1], doz = mm.z - vmx.campos[2];
                    const float len = sqrtf(dox * dox + doy * doy + doz * doz);
                    const ShDir sd(vp.D, dox / len, doy / len, doz / len);
                    const int ncoef = (vp.D + 1) * (vp.D + 1);
                    const float4* row = reinterpret_cast<const float4*>(shs) + (size_t)idx * 12;
#pragma unroll
                    for (int j = 0; j < 12; j++) {
                        const float4 v = row[j];
                        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int e = 4 * j + t, i = e / 3, ch = e % 3;
                            if (i < ncoef) sd.feed(i, f[t], shd[ch], shd[3 + ch], shd[6 + ch]);
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 9; k++) shd[k] = sh_dir[(size_t)sh_row * 9 + k];
                }
                float dRGB[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) dRGB[ch] = part[ch] * (((clamp_bits >> ch) & 1u) ? 0.f : 1.f);
                view_dir_terms[3 * (size_t)idx] = shd[0] * dRGB[0] + shd[1] * dRGB[1] + shd[2] * dRGB[2];
                view_dir_terms[3 * (size_t)idx + 1] = shd[3] * dRGB[0] + shd[4] * dRGB[1] + shd[5] * dRGB[2];
                view_dir_terms[3 * (size_t)idx + 2] = shd[6] * dRGB[0] + shd[7] * dRGB[1] + shd[8] * dRGB[2];
            }
        }
        if (valid) {
#pragma unroll
            for (int c = 0; c < FRG_SLOT_FLOATS; c++) sums[(size_t)idx * FRG_SLOT_FLOATS + c] = part[c];
            if (dL_dcolor) {
                const bool masked = shs && !dL_dsh;
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    dL_dcolor[3 * idx + ch] = (masked && !(visible && !((clamp_bits >> ch) & 1u))) ? 0.0f : part[ch];
            }
        }
        continue;
    }
    if (flags & FRG_PBW_FROM_SUMS) {
#pragma unroll
        for (int c = 0; c < FRG_SLOT_FLOATS; c++) part[c] = valid ? sums[(size_t)idx * FRG_SLOT_FLOATS + c] : 0.0f;
    }
    // LIVE Gaussians: those whose slot sums are not all zero.  At C3 only one visible Gaussian in seven is reached by
    // a pixel before its tiles saturate (370 000 of 2.5 M); for the others every term below is a product with these
    // zeros, so every gradient row is zero: they skip the loads (sh_dir 36 B, mean 12, scale 12, quaternion 16) and the
    // arithmetic, and write their zero rows.  (The rows they wrote before were +-0 from the same products.)
    bool has_grad = false;
#pragma unroll
    for (int c = 0; c < FRG_SLOT_FLOATS; c++) has_grad |= part[c] != 0.0f;
    has_grad &= visible;
    if (ablate & 8) has_grad = false;   // TIMING EXPERIMENT ONLY: no per-Gaussian mathematics, zero rows
    // row_live (frg_backward_args, optional): the caller wants to know which Gaussians have a gradient INSTEAD of their zero
    // rows -- one byte per Gaussian is written, and only the rows of the marked ones (at C3 six rows in seven are zeros:
    // 0.74 of the 0.85 GB this kernel writes)
    const bool wr = valid && (!row_live || has_grad);
    if (row_live && valid) row_live[idx] = has_grad ? 1 : 0;
    // d(colour)/d(direction), left by the forward's SH pass (GeomState::sh_dir): the backward does not read the 192-byte SH rows
    // -- or not left (sh_layout bit 1, r05): then they are formed in section 4 below, from the SH rows of the Gaussians that
    // have a gradient only, in the forward's order of additions (the same bits either way)
    const bool dir_here = SH16 && shs && (*sh_layout & 2u);          // wave-uniform
    float shd[9];
#pragma unroll
    for (int k = 0; k < 9; k++) shd[k] = (has_grad && shs && !dir_here) ? sh_dir[(size_t)sh_row * 9 + k] : 0.0f;
    if (dir_here && has_grad) {
        // shd[3 a + ch] = sum_i dbasis_i/d(axis a) * sh[i][ch], coefficient after coefficient as the forward's SH pass adds them
        // (ShDir::feed; preprocess.hip sh_pass): 192 bytes of this Gaussian's row, read by the Gaussians with a gradient only.
        // Here, in front of the covariance chain, the twelve requests in flight cost no registers beyond the kernel's peak.
        const float3 m = param_mean(means3D, raw, idx);
        const float dox = m.x - vmx.campos[0], doy = m.y - vmx.campos[1], doz = m.z - vmx.campos[2];   // the expressions of section 4 (and of the forward)
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);
        const ShDir sd(vp.D, dox / len, doy / len, doz / len);
        const int ncoef = (vp.D + 1) * (vp.D + 1);
        const float4* row = reinterpret_cast<const float4*>(shs) + (size_t)idx * 12;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const float4 v = row[j];
            const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int e = 4 * j + t, i = e / 3, ch = e % 3;
                if (i < ncoef) sd.feed(i, f[t], shd[ch], shd[3 + ch], shd[6 + ch]);
            }
        }
    }
    // The blend backward stores pixel MOMENTS of v = G dL/dalpha per (tile, Gaussian): sum v dx, v dy, v dx^2,
    // v dx dy, v dy^2, v.  The map to the reference's terms (backward.cu:536-554: dL/dG = o dL/dalpha,
    // dG/d(delta) = -G (a dx + b dy, c dy + b dx), d(delta)/d(NDC) = (W/2, H/2)) is linear with per-GAUSSIAN
    // coefficients, so it is applied here, once per Gaussian after the sum over its tiles, instead of once per
    // (tile, Gaussian) instance in the blend kernel.
    if (has_grad) {
        const float4 kc = conic_opacity[FRG_REC * idx];
        const float o = kc.w, m3 = part[3], m4 = part[4];
        part[3] = -o * (kc.x * m3 + kc.y * m4) * (0.5f * vp.W);
        part[4] = -o * (kc.z * m4 + kc.y * m3) * (0.5f * vp.H);
        part[5] = -0.5f * o * part[5];
        part[6] = -0.5f * o * part[6];
        part[7] = -0.5f * o * part[7];
    }

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float3 mean = make_float3(0.f, 0.f, 0.f);
    if (has_grad) {
        mean = param_mean(means3D, raw, idx);
        // ---- 2. computeCov2DCUDA (backward.cu:144-274) ----
        float cov[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov[i] = cov3D_precomp[6 * idx + i];
        } else {
            const float3 sc = param_scale(scales, raw, idx);
            const float4 q = param_rot(rotations, raw, idx);
            cov3d_from_scale_rot(sc, vp.scale_modifier, q, cov);  // recomputed, bit-identical to forward
        }
        const float dLc0 = part[5], dLc1 = part[6], dLc3 = part[7];
        const Ewa e = ewa_setup(mean, vp.focal_x, vp.focal_y, vp.tan_fovx, vp.tan_fovy, vmx.view);
        float a, b, c;
        ewa_cov2d(e, cov, a, b, c);
        a += 0.3f; c += 0.3f;
        const float denom = a * c - b * b;
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
#define W_(k, rr) vmx.view[4 * (rr) + (k)]
        const float dL_dJ00 = W_(0, 0) * dL_dT00 + W_(0, 1) * dL_dT01 + W_(0, 2) * dL_dT02;
        const float dL_dJ02 = W_(2, 0) * dL_dT00 + W_(2, 1) * dL_dT01 + W_(2, 2) * dL_dT02;
        const float dL_dJ11 = W_(1, 0) * dL_dT10 + W_(1, 1) * dL_dT11 + W_(1, 2) * dL_dT12;
        const float dL_dJ12 = W_(2, 0) * dL_dT10 + W_(2, 1) * dL_dT11 + W_(2, 2) * dL_dT12;
#undef W_
        const float h_x = vp.focal_x, h_y = vp.focal_y;
        const float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dL_dtx = e.xmul * -h_x * tz2 * dL_dJ02;
        const float dL_dty = e.ymul * -h_y * tz2 * dL_dJ12;
        const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * e.t[0]) * tz3 * dL_dJ02 + (2 * h_y * e.t[1]) * tz3 * dL_dJ12;
        const float* vm = vmx.view;
        dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
        // ---- 3. projection path (backward.cu:367-387) ----
        const float* proj = vmx.proj;
        const float4 m_hom = xform44(mean, proj);
        const float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        const float g2x = part[3], g2y = part[4];
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    }
    // screen-space outputs (also returned to the caller: viewspace gradients)
    if (wr) {
        dL_dmean2D[3 * idx] = part[3]; dL_dmean2D[3 * idx + 1] = part[4]; dL_dmean2D[3 * idx + 2] = 0.0f;
        if (dL_dconic) *reinterpret_cast<float4*>(dL_dconic + 4 * idx) = make_float4(part[5], part[6], 0.0f, part[7]);
        // raw mode: d sigmoid = o (1 - o)
        dL_dopacity[idx] = (raw.raw_opacity && has_grad) ? part[8] * ((1.0f - conic_opacity[FRG_REC * idx].w) * conic_opacity[FRG_REC * idx].w) : part[8];
        // with shs given and dL_dsh == nullptr the caller wants the factor of the SH gradient instead
        // (the clamp-masked colour gradient, stored below): see frg_backward in the header
        if (dL_dcolor && !(shs && !dL_dsh) && !(flags & FRG_PBW_FROM_SUMS)) { dL_dcolor[3 * idx] = part[0]; dL_dcolor[3 * idx + 1] = part[1]; dL_dcolor[3 * idx + 2] = part[2]; }
    }

    // ---- 4. SH path (backward.cu:20-139) ----
    if (shs && !(ablate & 2)) {
        // dL/dsh[i][ch] = wgt[i] * dRGB[ch]; the view-direction gradient needs
        // d(colour)/d(dir) = sum_i dbasis_i/d(dir) * sh[i]: the forward left it in sh_dir (shd)
        float wgt[16], dRGB[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; i++) wgt[i] = 0.0f;
        const int M = vp.M;
        const int deg = vp.D;
        float x = 0.f, y = 0.f, z = 1.f, dox = 0.f, doy = 0.f, doz = 1.f;
        if (has_grad) {
            dox = mean.x - vmx.campos[0]; doy = mean.y - vmx.campos[1]; doz = mean.z - vmx.campos[2];
            const float len = sqrtf(dox * dox + doy * doy + doz * doz);
            x = dox / len; y = doy / len; z = doz / len;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = part[ch] * (((clamp_bits >> ch) & 1u) ? 0.f : 1.f);
        }
        if (!dL_dsh && wr && !(flags & FRG_PBW_FROM_SUMS)) { dL_dcolor[3 * idx] = dRGB[0]; dL_dcolor[3 * idx + 1] = dRGB[1]; dL_dcolor[3 * idx + 2] = dRGB[2]; }
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        if (has_grad) {
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
        }
        if (has_grad) {
            const float dd0 = shd[0] * dRGB[0] + shd[1] * dRGB[1] + shd[2] * dRGB[2];
            const float dd1 = shd[3] * dRGB[0] + shd[4] * dRGB[1] + shd[5] * dRGB[2];
            const float dd2 = shd[6] * dRGB[0] + shd[7] * dRGB[1] + shd[8] * dRGB[2];
            // auxiliary.h:107-117 dnormvdv
            const float sum2 = dox * dox + doy * doy + doz * doz;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((+sum2 - dox * dox) * dd0 - doy * dox * dd1 - doz * dox * dd2) * invsum32;
            dmean[1] += (-dox * doy * dd0 + (sum2 - doy * doy) * dd1 - doz * doy * dd2) * invsum32;
            dmean[2] += (-dox * doz * dd0 - doy * doz * dd1 + (sum2 - doz * doz) * dd2) * invsum32;
        }
        // gradient rows: coefficients above the active degree and culled Gaussians are zero.
        // dL_dsh == nullptr: the caller rebuilds the (summed) SH gradient from the colour gradient
        // (view_exchange.hip) and the 192 B/Gaussian row is not materialised per view.
        if (!dL_dsh) {
        } else if (SH16) {
            float4* dst = reinterpret_cast<float4*>(dL_dsh) + (size_t)idx0 * 12;
            const int nvalid = min(64, P - idx0);
            // rows to write: all of the wave's -- or, with row_live, those of its Gaussians with a gradient
            const uint64_t wmask = row_live ? __ballot(has_grad) : ~0ull;
#pragma unroll 1
            for (int h = 0; h < 64 / BWD_SUB; h++) {
                if (((wmask >> (h * BWD_SUB)) & ((1ull << BWD_SUB) - 1ull)) == 0ull) continue;     // wave-uniform: nothing of this sub-batch is written
                if ((lane / BWD_SUB) == h) {
#pragma unroll
                    for (int j = 0; j < 12; j++)
                        shbuf[(lane % BWD_SUB) * BWD_ROW_F4 + j] =
                            make_float4(wgt[(4 * j) / 3] * dRGB[(4 * j) % 3], wgt[(4 * j + 1) / 3] * dRGB[(4 * j + 1) % 3],
                                        wgt[(4 * j + 2) / 3] * dRGB[(4 * j + 2) % 3], wgt[(4 * j + 3) / 3] * dRGB[(4 * j + 3) % 3]);
                }
                wave_fence();
#pragma unroll
                for (int k = 0; k < BWD_SUB * 12 / 64; k++) {
                    const int f = k * 64 + lane, gl = f / 12, j = f - gl * 12;
                    if (h * BWD_SUB + gl < nvalid && ((wmask >> (h * BWD_SUB + gl)) & 1ull)) {
                        // written once, read by nobody in this op: past the L2 (576 MB per view that would push the slots out)
                        typedef float nt_f4 __attribute__((ext_vector_type(4)));
                        const float4 v = shbuf[gl * BWD_ROW_F4 + j];
                        __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4*>(dst + (size_t)h * BWD_SUB * 12 + f));
                    }
                }
                wave_fence();
            }
        } else if (wr) {
            float* o = dL_dsh + (size_t)idx * M * 3;
            const int n = min(M, 16) * 3;
#pragma unroll
            for (int i = 0; i < 48; i++)
                if (i < n) o[i] = wgt[i / 3] * dRGB[i % 3];
            for (int i = 48; i < M * 3; i++) o[i] = 0.0f;
        }
    }
    if (wr) {
        dL_dmean3D[3 * idx] = dmean[0]; dL_dmean3D[3 * idx + 1] = dmean[1]; dL_dmean3D[3 * idx + 2] = dmean[2];
#pragma unroll
        for (int i = 0; i < 6; i++) if (dL_dcov3D) dL_dcov3D[6 * idx + i] = dcov[i];
    }
    // shell-bound centres (frosting_model.py:707-724): mean = sum_k w_k v_k, w = softmax(logits) or relu(x) / sum relu(x).
    //   dL/dlogit_k = w_k (g_k - sum_j w_j g_j), g_k = v_k . dL/dmean          (softmax Jacobian)
    //   dL/dx_k     = [x_k > 0] (g_k - sum_j w_j g_j) / sum relu(x)            (relu + renormalise)
    //   dL/dv_k    += w_k dL/dmean                                              (learnable shell, learn_shell = True)
    if (raw.shell_logits && wr) {
        float gl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (has_grad) {
            float w[6], gk[6];
            raw_bary6(raw, idx, w);
            const size_t cell = (size_t)raw.shell_cells[idx];
            const float* v = raw.shell_verts + 18 * cell;
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                gk[k] = v[3 * k] * dmean[0] + v[3 * k + 1] * dmean[1] + v[3 * k + 2] * dmean[2];
                dot += w[k] * gk[k];
            }
#pragma unroll
            for (int k = 0; k < 6; k++) gl[k] = w[k] * (gk[k] - dot);
            if (raw.bary_mode == 1) {
                const float* x = raw.shell_logits + 6 * (size_t)idx;
                float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < 6; k++) ssum += fmaxf(x[k], 0.0f);
#pragma unroll
                for (int k = 0; k < 6; k++) gl[k] = x[k] > 0.0f ? (gk[k] - dot) / ssum : 0.0f;
            }
            if (dL_dshell_verts) {
                // several Gaussians share a cell: float atomics (the one place of this path whose summation order is
                // not fixed; the reference's autograd index_add has the same property)
#pragma unroll
                for (int k = 0; k < 6; k++)
#pragma unroll
                    for (int c = 0; c < 3; c++) atomicAdd(dL_dshell_verts + 18 * cell + 3 * k + c, w[k] * dmean[c]);
            }
        }
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dshell_logits[6 * (size_t)idx + k] = gl[k];
    }

    // ---- 5. cov3D -> scale, quaternion (backward.cu:278-341) ----
    if ((scales || raw.raw_scale) && wr) {
        float ds[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0};
        if (has_grad) {
            const float3 sc = param_scale(scales, raw, idx);
            const float4 q = param_rot(rotations, raw, idx);
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            const Rot3 R = quat_to_rot(q);
            const float s[3] = {vp.scale_modifier * sc.x, vp.scale_modifier * sc.y, vp.scale_modifier * sc.z};
            float Mm[3][3];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) Mm[c][rr] = s[rr] * R.c[c][rr];
            const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                    {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            float dMt[3][3];  // dMt[c][r] = dM[r][c], dM = (2 M) dSigma
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++)
                    dMt[rr][c] = (Mm[0][rr] * 2.0f) * dS[c][0] + (Mm[1][rr] * 2.0f) * dS[c][1] + (Mm[2][rr] * 2.0f) * dS[c][2];
#pragma unroll
            for (int c = 0; c < 3; c++) ds[c] = R.c[0][c] * dMt[c][0] + R.c[1][c] * dMt[c][1] + R.c[2][c] * dMt[c][2];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) dMt[c][rr] *= s[c];
            dq[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
            dq[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
            dq[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
            dq[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
        }
        if (has_grad && raw.raw_scale) {          // d exp = exp
            const float3 sc = param_scale(scales, raw, idx);
            ds[0] *= sc.x; ds[1] *= sc.y; ds[2] *= sc.z;
        }
        if (has_grad && raw.raw_rot) {            // y = x / max(|x|, eps): dx = (g - y (y . g)) / max(|x|, eps)
            const float4 x = make_float4(raw.raw_rot[4 * idx], raw.raw_rot[4 * idx + 1], raw.raw_rot[4 * idx + 2], raw.raw_rot[4 * idx + 3]);
            const float nrm = sqrtf(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
            const float inv = 1.0f / fmaxf(nrm, 1e-12f);
            const float4 y = make_float4(x.x * inv, x.y * inv, x.z * inv, x.w * inv);
            const float d = nrm > 1e-12f ? (y.x * dq[0] + y.y * dq[1] + y.z * dq[2] + y.w * dq[3]) : 0.0f;
            dq[0] = (dq[0] - y.x * d) * inv; dq[1] = (dq[1] - y.y * d) * inv;
            dq[2] = (dq[2] - y.z * d) * inv; dq[3] = (dq[3] - y.w * d) * inv;
        }
        dL_dscale[3 * idx] = ds[0]; dL_dscale[3 * idx + 1] = ds[1]; dL_dscale[3 * idx + 2] = ds[2];
        *reinterpret_cast<float4*>(dL_drot + 4 * idx) = make_float4(dq[0], dq[1], dq[2], dq[3]);
    }
  } while (HEAVY && (heavy_item += gridDim.x) < heavy[0]);
}

hipError_t launch_preprocess_bwd(int P, const ViewParams& vp, const FwdInputs& in, const int* radii, const GeomState& g,
                                 const ImageState& img, const float* slots, const BwdOutputs& o, int ablate, int flags,
                                 bool heavy_only, hipStream_t s, float* sums, unsigned long long* live_masks, float* view_dir_terms,
                                 int range_first, int range_count)
{
    // (range: the plain kernel over Gaussians [range_first, range_first + range_count) only, range_first a multiple of 256)
    const int first_block = range_count > 0 ? range_first / BWD_THREADS : 0;
    const dim3 grid(((range_count > 0 ? range_count : P) + BWD_THREADS - 1) / BWD_THREADS), block(BWD_THREADS);
    // float4-streamed SH needs the reference's usual layout: 16 coefficients, 16-byte aligned rows
    const bool sh16 = in.shs && vp.M == 16 && (reinterpret_cast<uintptr_t>(in.shs) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(o.dL_dsh) % 16 == 0);
#define FRG_PBW(S16, HV, GRID, BLOCK)                                                                                  \
    hipLaunchKernelGGL((preprocess_bwd_kernel<S16, HV>), GRID, BLOCK, 0, s, P, vp, in.viewmatrix, in.projmatrix,              \
                       in.cam_pos, in.means3D, radii, in.shs, in.scales, in.rotations, in.cov3D_precomp, g.xydr,          \
                       g.rgb_clamped, g.conic_opacity, g.point_offsets, img.cutoff, img.counters, slots, o.dL_dmean2D,     \
                       o.dL_dconic, o.dL_dopacity, o.dL_dcolor, o.dL_dmean3D, o.dL_dcov3D, o.dL_dsh, o.dL_dscale, o.dL_drot, ablate,       \
                       in.raw, o.dL_dshell_logits, o.dL_dshell_verts, g.sh_dir, flags, g.heavy_waves, g.sh_layout, sums, o.row_live, live_masks, view_dir_terms, first_block)
    // the listed waves (usually none: the workgroups read the count and leave)
    const dim3 hgrid(256), hblock(BWD_HEAVY_WAVES * 64);
    if (heavy_only) { if (sh16) FRG_PBW(true, true, hgrid, hblock); else FRG_PBW(false, true, hgrid, hblock); }
    else { if (sh16) FRG_PBW(true, false, grid, block); else FRG_PBW(false, false, grid, block); }
#undef FRG_PBW
    return hipGetLastError();
}

}  // namespace frg
