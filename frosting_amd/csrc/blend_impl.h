// Front-to-back alpha compositing, forward and backward (the reference's K6 / K7:
// forward.cu:261-374, backward.cu:399-557), re-designed for CDNA4 wave64:
//
//   * FORWARD: one WAVE per 8x8 QUADRANT of a 16x16 tile (a 256-thread workgroup = the four
//     quadrants of one tile; its waves never synchronise with each other), one pixel per
//     lane: 45 VGPRs, so 8 waves per SIMD hide the LDS / transcendental latency of the
//     per-Gaussian dependent chain, and a quadrant that saturates retires on its own;
//   * BACKWARD: one wave per SEGMENT of a tile's processed list prefix (256 / 512 entries by frame; the forward
//     leaves the pixels' state at the boundaries, so a segment starts anywhere), four pixels per lane
//     (one per quadrant): the nine gradient components of a (tile, Gaussian) pair are first summed over
//     the lane's pixels, so the cross-lane reduction -- the single most expensive step -- is paid once
//     per tile instance instead of once per quadrant instance (measured: 0.72 vs 0.79 ms);
//   * instances are staged 64 at a time through wave-private LDS from 16-byte
//     per-Gaussian records (three global_load_dwordx4 gathers per instance);
//   * EXACT QUADRANT CULLING (quadrant_hit, frg_common.h): while staging, the lane that
//     fetched a Gaussian bounds its falloff over each quadrant in closed form and the
//     wave compacts the staged list by ballot.  The reference's tile lists are 3-sigma
//     squares: more than half of the (quadrant, Gaussian) pairs never reach alpha >= 1/255,
//     and they are dropped while every pixel keeps exactly its value;
//   * the backward reduction takes the 27 partial sums of three instances through an LDS
//     matrix (one column per lane) and stores each (tile, Gaussian) gradient ONCE, without
//     atomics, in the instance's Gaussian-major slot; the per-Gaussian backward kernel sums the slots in a
//     fixed order.  The reference issues 9 global float atomics per (pixel, Gaussian)
//     pair (backward.cu:523,545-554) and is not reproducible run to run.
//
// EXACT=true : IEEE operation order of the reference, no contraction, accurate expf
//              (bit-identical image to the reference built with -ffp-contract=off).
// EXACT=false: FMA contraction and native exp2 (default product path).
// (template bodies; instantiated by blend_exact.hip / blend_fast.hip, which fix the
// floating-point contraction mode for everything below)
#pragma once
#include "frg_common.h"
#include "sort_lds.h"
#include <algorithm>

namespace frg {

template <bool EXACT>
struct BlendMath;

template <>
struct BlendMath<true> {
    static __device__ __forceinline__ float power(float x, float y, float4 co, float px, float py, float& dx, float& dy)
    {
        dx = x - px; dy = y - py;
        return -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
    }
    static __device__ __forceinline__ float expo(float p) { return expf(p); }
    static __device__ __forceinline__ float mul3(float a, float b, float c) { return a * b * c; }
    static __device__ __forceinline__ float recip(float x) { return 1.0f / x; }
    static __device__ __forceinline__ float mad(float a, float b, float c) { return a * b + c; }   // two roundings (this TU: contraction off)
    // what the staging lane leaves in LDS for the per-pixel loop: the conic as the reference holds it
    static __device__ __forceinline__ float4 stage(float4 co) { return co; }
    // forward.cu:347,358-362: test_T = T (1 - alpha); C[ch] += features[ch] * alpha * T
    static __device__ __forceinline__ float attenuate(float T, float alpha) { return T * (1 - alpha); }
    static __device__ __forceinline__ void accumulate(float4 c, float alpha, float T, float& C0, float& C1, float& C2)
    {
        C0 = mad(c.x * alpha, T, C0);
        C1 = mad(c.y * alpha, T, C1);
        C2 = mad(c.z * alpha, T, C2);
    }
};

template <>
struct BlendMath<false> {
    // Every fused multiply-add of the fast path is written out (the TU is compiled with contraction off): what
    // gets fused must not depend on where the compiler unrolled or inlined a copy of the code, or the same
    // (pixel, Gaussian) pair would round differently at different positions of a tile list.
    // The staging lane folds the -1/2 of the exponent and the log2(e) of exp -> exp2 into the conic once per
    // Gaussian (stage()); per pixel: power' = dx (a' dx + b' dy) + c' dy^2 in log2 units, G = exp2(power').
    // Seven instructions and one v_exp instead of ten and one; `power > 0` keeps its sign.
    // (round 5's one-ingredient-at-a-time A/B switches of this arithmetic: profiles/r05_blend_ab_switches.diff)
    static __device__ __forceinline__ float4 stage(float4 co)
    {
        const float l2e = 1.4426950408889634f;
        return make_float4(-0.5f * l2e * co.x, -l2e * co.y, -0.5f * l2e * co.z, co.w);
    }
    static __device__ __forceinline__ float power(float x, float y, float4 sc, float px, float py, float& dx, float& dy)
    {
        dx = x - px; dy = y - py;
        const float t = __builtin_fmaf(sc.x, dx, sc.y * dy);
        return __builtin_fmaf(t, dx, (sc.z * dy) * dy);
    }
    static __device__ __forceinline__ float expo(float p) { return __builtin_amdgcn_exp2f(p); }
    static __device__ __forceinline__ float mul3(float a, float b, float c) { return a * b * c; }
    static __device__ __forceinline__ float recip(float x) { return __builtin_amdgcn_rcpf(x); }  // v_rcp_f32, 1 ulp
    static __device__ __forceinline__ float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    // T (1 - alpha) as one FMA; the weight alpha T once per pixel and one FMA per channel (four instructions
    // instead of six, one instead of two: 3 of the forward blend's 24 vector instructions per list entry)
    static __device__ __forceinline__ float attenuate(float T, float alpha) { return __builtin_fmaf(-alpha, T, T); }
    static __device__ __forceinline__ void accumulate(float4 c, float alpha, float T, float& C0, float& C1, float& C2)
    {
        const float w = alpha * T;
        C0 = __builtin_fmaf(c.x, w, C0);
        C1 = __builtin_fmaf(c.y, w, C1);
        C2 = __builtin_fmaf(c.z, w, C2);
    }
};

#define BLEND_THREADS 256   // 4 waves = the 4 quadrants of one tile

// 64-bit lane mask of a predicate, straight from the compare (HIP's __ballot(int) first materialises the
// predicate as 0 / 1 in a VGPR and compares it again)
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ int lanes_before(uint64_t mask, int lane) { return __popcll(mask & ((1ull << lane) - 1ull)); }

// wave-private LDS hand-off (no workgroup barrier anywhere in these kernels)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// PREFETCH (the default, option "fwd_prefetch"): the records of round r + 1 are requested before round r is processed
// -- 52 instead of 41 VGPRs, still eight waves per SIMD.  A frame of few, long tile lists is bound by the per-round
// gather latency of its longest tiles' waves (C4: 0.321 -> 0.282 ms); with thousands of waves in flight the other
// waves hide most of it (C3: 0.2135 -> 0.2025).  Same loads, same arithmetic: every output bit-identical.  (Round 2
// measured a register double-buffering of the three separate arrays slower; with 48-byte records it pays.)
// FUSED (small frames, round 6): a tile whose list has at most FRG_FUSED_SORT_CAP entries is SORTED HERE, by the tile's four waves
// together (sort_lds.h: the tile sort's own passes, 2 entries per thread), from the scatter's unsorted pairs -- the host skips the
// tile sort's launch for that size class; the sorted indices stay in LDS for the walk (and go to point_list for the record and
// the backward).  A frame of 100 k Gaussians (C2: 2 500 lists of 140 entries) is six launches of 5 - 30 us, each a chain of
// dependent memory round trips: this takes one launch boundary and the point_list round trip out of it.
#define FRG_FUSED_SORT_CAP 512
template <bool EXACT, bool PREFETCH = false, bool FUSED = false, int UNROLL = 4>
__global__ void __launch_bounds__(BLEND_THREADS)
blend_fwd_kernel(int T, int gx, int W, int H, const uint2* __restrict__ ranges,
                 uint32_t* point_list /* FUSED: written here for the short lists */, const float4* __restrict__ xydr,
                 const float4* __restrict__ conic_opacity, const float4* __restrict__ rgb_clamped,
                 const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                 float* __restrict__ out_color, uint32_t* __restrict__ tile_work, float4* __restrict__ ckpt,
                 float4* __restrict__ final_C, const uint32_t* __restrict__ class_tiles, const uint32_t* __restrict__ class_count,
                 int seg_log, uint32_t* __restrict__ bwd_cnt, uint32_t* __restrict__ bwd_last, uint32_t cap_b,
                 uint2* __restrict__ bwd_full, uint32_t cap_a, uint2* __restrict__ cutoff, Counters* __restrict__ counters,
                 const uint2* __restrict__ pairs)
{
    using M = BlendMath<EXACT>;
    const int SEG = 1 << seg_log;      // entries per segment of the backward blend (frg_common.h: bwd_seg_log)
    // how deep the tile was walked, over its four quadrant waves: the last of them to finish lists the tile's backward items
    __shared__ uint32_t s_deep, s_done;
    if (threadIdx.x == 0) { s_deep = 0u; s_done = 0u; }
    __syncthreads();                   // the only workgroup barrier of the kernel: the four waves start together anyway
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counters->bwd_seg_log = (uint32_t)seg_log;
        counters->fwd_flags = FRG_FWD_STAMPED | (EXACT ? FRG_FWD_EXACT : 0u) | (ckpt ? 0u : FRG_FWD_ONLY);
    }
    int tile;
    if (class_tiles) {
        // The tiles LONGEST LIST FIRST (the sort's size classes, longest class first, eight descending buckets inside a
        // class, the empty tiles last): a tile's four waves are a sequential walk, and in tile order the long ones start
        // whenever their turn comes -- the kernel then ends with them.  Same box, band-by-band order (rounds 1-3, which kept
        // neighbouring tiles on one XCD's L2) against this: C3 0.208 -> 0.198 ms, C4 0.280 -> 0.253, C2 0.040 -> 0.031,
        // clustered scene 0.234 -> 0.177.  (option fwd_order = 0: the band order)
        int k = (int)blockIdx.x;
        if (k >= T) return;
        tile = -1;
#pragma unroll
        for (int c = FRG_SORT_CLASSES - 1; c >= 0; c--) {
            const int n = (int)class_count[c];
            if (tile < 0 && k < n) tile = (int)class_tiles[(size_t)c * T + k];
            if (tile < 0) k -= n;
        }
        if (tile < 0) tile = (int)class_tiles[(size_t)FRG_SORT_CLASSES * T + k];
    } else tile = xcd_tile_of_block(blockIdx.x, T);
    if (tile < 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);

    __shared__ float4 s_a_all[4][64];    // x, y, -, contributor (1-based list position)
    __shared__ float4 s_co_all[4][64];   // conic a, b, c, opacity
    __shared__ float4 s_rgb_all[4][64];
    float4* s_a = s_a_all[q];
    float4* s_co = s_co_all[q];
    float4* s_rgb = s_rgb_all[q];

    // FUSED: the short list sorted by the four waves; its Gaussian indices then come from LDS
    __shared__ uint2 s_sorted[FUSED ? FRG_FUSED_SORT_CAP : 1];
    __shared__ uint32_t s_whist[FUSED ? 4 * 256 : 1];
    __shared__ uint32_t s_sscratch[FUSED ? FRG_SORT_SCRATCH_WORDS : 1];
    const bool in_lds = FUSED && n <= FRG_FUSED_SORT_CAP;          // workgroup-uniform
    if (in_lds && n > 0) {
        int begin, end;
        wave_strip<4>(n, q, begin, end);
        uint2 e[FRG_FUSED_SORT_CAP / BLEND_THREADS];
#pragma unroll
        for (int it = 0; it < FRG_FUSED_SORT_CAP / BLEND_THREADS; it++) {
            const int i = begin + it * 64 + lane;
            e[it] = i < end ? load_pair_stream(pairs + rg.x + i) : make_uint2(0u, 0u);
        }
        sort_block_lds<4, FRG_FUSED_SORT_CAP / BLEND_THREADS>(e, n, begin, end, s_sorted, s_whist, s_sscratch);
        for (int i = threadIdx.x; i < n; i += BLEND_THREADS) point_list[rg.x + i] = s_sorted[i].y;
    }

    const int qx0 = tx * FRG_TILE + (q & 1) * 8, qy0 = ty * FRG_TILE + (q >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const bool inside = px < W && py < H;
    // 1 while the pixel still blends, 0 once it has stopped (forward.cu:347-351) or lies outside the image.  A FLOAT
    // factor of alpha rather than a boolean in the three tests: a finished pixel then fails the alpha test by itself
    // (alpha * 1 == alpha exactly), and the per-lane state costs one v_cndmask instead of scalar mask arithmetic --
    // rocprofv3 counted 0.73 scalar instructions per vector instruction in this kernel, most of them the exec-mask
    // bookkeeping of a loop that every lane left on its own (`for (...; !done && ...)`), and the CU's four SIMDs
    // share one scalar unit.  The trip count below is the wave's; the wave leaves when its last pixel is done.
    float alive = inside ? 1.0f : 0.0f;
    float Tr = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0;

    // The records of round r + 1 are requested while round r is processed -- and the list entries (Gaussian indices) of round
    // r + 2 with them (r05): index -> record is a chain of two memory round trips, and a frame whose tail is a few long lists
    // (C4's limb) pays one such chain per round of 64 entries; with the indices a round further ahead it is one trip.
    float4 a_n = make_float4(0.f, 0.f, 0.f, 0.f), co_n = a_n, col_n = a_n;
    uint32_t id_n = 0;                 // this lane's list entry of the round after the one whose records are in flight
    auto fetch_id = [&](int base) {          // unconditional loads at clamped positions (see blend_bwd_kernel)
        id_n = in_lds ? s_sorted[min(base + lane, n - 1)].y : point_list[rg.x + min(base + lane, n - 1)];
    };
    auto fetch = [&](uint32_t id) {
        a_n = xydr[FRG_REC * id];
        co_n = conic_opacity[FRG_REC * id];
        col_n = rgb_clamped[FRG_REC * id];
    };
    if (PREFETCH && n > 0) { fetch_id(0); fetch(id_n); if (n > 64) fetch_id(64); }
    // The list is walked segment by segment (SEG entries: the work items of the backward blend).  At every
    // boundary the quadrant crosses, the state of its pixels BEFORE the segment's first entry is left for the backward's
    // item of the segment that ends there (a pixel that has stopped leaves stale values nobody reads: its last contributor
    // lies in front of the boundary): 1 KB per quadrant and boundary, contiguous.  The inner loop is the loop of rounds 1-3.
    bool saturated = false;
    for (int sbase = 0; sbase < n && !saturated; sbase += SEG) {
    if (sbase != 0) {
        if (wave_ballot(alive != 0.0f) == 0ull) break;
        if (ckpt) ckpt[((size_t)(rg.x >> seg_log) + (size_t)(sbase >> seg_log)) * FRG_TILE_PIX + q * 64 + lane] = make_float4(Tr, C0, C1, C2);
    }
    const int send = min(n, sbase + SEG);
    for (int base = sbase; base < send; base += 64) {
        if (wave_ballot(alive != 0.0f) == 0ull) { saturated = true; break; }   // this quadrant is saturated
        const int cnt = min(64, n - base);
        bool hit = false;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), co = a, col = a;
        if (PREFETCH) {
            a = a_n; co = co_n; col = col_n;
            if (base + 64 < n) {                          // wave-uniform
                fetch(id_n);
                if (base + 128 < n) fetch_id(base + 128);
            }
            hit = lane < cnt && quadrant_hit(a.x, a.y, co, qx0, qy0);
        } else if (lane < cnt) {
            const uint32_t id = in_lds ? s_sorted[base + lane].y : point_list[rg.x + base + lane];
            a = xydr[FRG_REC * id];
            co = conic_opacity[FRG_REC * id];
            col = rgb_clamped[FRG_REC * id];
            hit = quadrant_hit(a.x, a.y, co, qx0, qy0);
        }
        const uint64_t keep = wave_ballot(hit);
        const int nkeep = __popcll(keep);
        wave_lds_sync();                          // previous round's readers are done
        if (hit) {
            const int d = lanes_before(keep, lane);
            s_a[d] = make_float4(a.x, a.y, 0.f, __uint_as_float((uint32_t)(base + lane + 1)));
            s_co[d] = M::stage(co);
            s_rgb[d] = col;
        }
        wave_lds_sync();
        // UNROLL entries per trip: the falloff of an entry (quadratic form, v_exp, min) does not depend on the pixel's
        // state -- only T (1 - alpha), the stop test and the colour update do -- so the alphas of a group are computed side by
        // side and the sequential part of a pixel's chain shrinks to three dependent operations per entry.  Same operations on
        // the same values in the same order: every output bit-identical whatever UNROLL.  4 by default (same box, C4 forward
        // blend: 1 -> 0.256 ms, 2 -> 0.249, 4 -> 0.245; C3 and C2 unchanged: profiles/r05_ab_fwd_unroll.log); 8 (r06) for frames
        // whose work sits in a few long lists -- C4's limb: the launch is stall-bound inside every wave at < 5 waves per SIMD,
        // 0.240 -> 0.222 ms -- and NOT for C3, whose walk is bound by instruction issue (0.203 -> 0.243 ms: the slots behind a
        // round's last entry are executed as no-ops, and the registers cost occupancy): the launcher chooses per frame
        // (profiles/r06_ab_fwd_unroll8.log).
        for (int j = 0; j < nkeep; j += UNROLL) {
            float al[UNROLL], pw[UNROLL], cw[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const bool there = j + u < nkeep;            // wave-uniform; a slot beyond the round's entries is a no-op (alpha 0)
                const float4 ga = s_a[min(j + u, 63)];
                const float4 gco = s_co[min(j + u, 63)];
                float dx, dy;
                const float power = M::power(ga.x, ga.y, gco, pxf, pyf, dx, dy);
                pw[u] = there ? power : 0.0f;
                al[u] = there ? fminf(0.99f, gco.w * M::expo(power)) : 0.0f;
                cw[u] = ga.w;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const float alpha = al[u];
                const bool keep_px = !(pw[u] > 0.0f) & !(alpha * alive < 1.0f / 255.0f);
                const float test_T = M::attenuate(Tr, alpha);
                const bool stop = keep_px & (test_T < 0.0001f);
                alive = stop ? 0.0f : alive;
                if (keep_px & !stop) {
                    const float4 gc = s_rgb[min(j + u, 63)];
                    M::accumulate(gc, alpha, Tr, C0, C1, C2);
                    Tr = test_T;
                    last = __float_as_uint(cw[u]);
                }
            }
            if (wave_ballot(alive != 0.0f) == 0ull) break;   // wave-uniform
        }
    }
    }

    if (inside) {
        const size_t plane = (size_t)H * W;
        const size_t pid = (size_t)py * W + px;
        final_T[pid] = Tr;
        n_contrib[pid] = last;
        out_color[pid] = M::mad(Tr, bg[0], C0);
        out_color[plane + pid] = M::mad(Tr, bg[1], C1);
        out_color[2 * plane + pid] = M::mad(Tr, bg[2], C2);
    }
    // forward_only (ckpt == nullptr, frg_forward_args::forward_only): the frame is complete here -- what follows is kept for a
    // backward: checkpoints above, final colours, walked depths, cutoff keys, work items.  (The item counters were cleared by
    // the binning: a backward called all the same finds no item and returns zero gradients.)
    if (!ckpt) return;
    // how deep this quadrant walked the tile's list
    uint32_t deepest = inside ? last : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) deepest = max(deepest, (uint32_t)__shfl_xor((int)deepest, d, 64));
    // a pixel whose last contributor lies behind a segment boundary: the backward needs the colour it ended with
    if (deepest > (uint32_t)SEG) final_C[(size_t)tile * FRG_TILE_PIX + q * 64 + lane] = make_float4(C0, C1, C2, 0.0f);
    // ---- the tile's work items for the backward blend (see "Work items" below) ----
    // The tile's processed prefix = the deepest walk of its four waves; the wave that finishes LAST (LDS counter) lists
    //   * its last segment in the length bucket of its XCD band (bwd_last, bwd_cnt[x][bucket]),
    //   * its full segments (tile, k) in the band's list (bwd_full, bwd_cnt[x][FRG_BWD_LEN_BUCKETS]),
    // with one relaxed device-scope atomic each: ~13 000 of them spread over the kernel's 0.2 ms at C3.  The order inside
    // a list is the order in which the tiles finished -- it differs from run to run, the backward's results do not
    // (every slot is written once, whoever processes its item: test_backward_blend_work_items_whatever_the_grid).
    // A tile in which nothing was blended gets its cutoff key cleared here (the per-Gaussian backward must not find
    // the key an earlier frame left for it).
    uint32_t prev = 0;
    if (lane == 0) {
        if (deepest) atomicMax(&s_deep, deepest);                // ds_max_u32
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        prev = atomicAdd(&s_done, 1u);                           // ds_add_rtn_u32
    }
    prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
    if (prev != BLEND_THREADS / 64 - 1) return;                 // wave-uniform: not the last wave of the tile
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t wk = __hip_atomic_load(&s_deep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0) tile_work[tile] = wk;
    if (wk == 0u) { if (lane == 0) cutoff[tile] = make_uint2(0u, 0u); return; }
    const int xb = xcd_of_tile(tile, T);
    const uint32_t nfull = (wk - 1u) >> seg_log;
    uint32_t at = 0;
    if (lane == 0) {
        const uint32_t bucket = (uint32_t)(FRG_BWD_LEN_BUCKETS - 1) - ((((wk - 1u) & (uint32_t)(SEG - 1)) * FRG_BWD_LEN_BUCKETS) >> seg_log);
        uint32_t* cnt = bwd_cnt + xb * (FRG_BWD_LEN_BUCKETS + 1);
        const uint32_t slot = __hip_atomic_fetch_add(&cnt[bucket], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (slot < cap_b) bwd_last[((size_t)xb * FRG_BWD_LEN_BUCKETS + bucket) * cap_b + slot] = (uint32_t)tile;
        if (nfull) at = __hip_atomic_fetch_add(&cnt[FRG_BWD_LEN_BUCKETS], nfull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (nfull) {
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        for (uint32_t sgm = (uint32_t)lane; sgm < nfull; sgm += 64u)
            if (at + sgm < cap_a) bwd_full[(size_t)xb * cap_a + at + sgm] = make_uint2((uint32_t)tile, sgm);
    }
}

// Work items of the backward blend.  A tile's processed prefix (tile_work[t] entries: up to the last contributor of its
// last pixel) is cut into SEGMENTS of 1 << seg_log entries; every segment is one work item of one wave.  Round 1-3 gave
// a whole tile to one wave (or, on frames with few active tiles, to four): 324..1077 entries at C3, but 10^3..5 10^3 at
// the limb of a shell seen from outside (C4) and 10^4..10^5 in a cluster -- the frame then waited for its deepest tile
// (C4: 0.58 ms for 2.9 M tile-entries, when C3's 4.3 M take 0.39).  A segment can start anywhere because the forward left the
// state there (BinningState::ckpt): the walk is a scan, and a scan can be restarted from a stored prefix.
// The items are listed by the FORWARD blend's tile workgroups as they finish (blend_fwd_kernel's tail; round 4 had a
// single-workgroup ordering kernel in front of the backward: 17 us of cold-code latency chain per step), per XCD band of
// tiles (neighbouring tiles share their Gaussians: same L2):
//   list A  the FULL segments (tile, k), k < nseg - 1: all the same length, longest items of the frame, pulled first;
//   list B  every active tile's LAST segment, in 32 buckets by decreasing length.
// Every backward wave derives from the 8 x 33 counters what round 4's kernel wrote into a header (BwdShares below).  The
// assignment is STATIC: XCD x's waves stride over its items (A then B, one index space) -- no work queue: 4096 waves
// pulling from eight cursors with device-scope atomics, which execute behind the XCDs' L2s one at a time per address,
// took 0.55 instead of 0.41 ms at C3.  What keeps the XCDs even is the pool: every XCD is given the same number of items
// (m = N / 8), an XCD with more than that donates its LAST (shortest) ones, one with fewer takes from the pool.
struct BwdShares {
    uint32_t cnt[FRG_NUM_XCD * (FRG_BWD_LEN_BUCKETS + 1)];   // the forward's counters, as loaded
    uint32_t own[FRG_NUM_XCD];      // items of the XCD's own band (A + B)
    uint32_t m[FRG_NUM_XCD];        // items its waves process
    uint32_t pool[FRG_NUM_XCD];     // own < m: its first index in the pool | own > m: the pool index of its first donated item
    // one wave: load the counters and derive the shares (s points to LDS)
    static __device__ __forceinline__ void build(BwdShares* s, const uint32_t* __restrict__ bwd_cnt, int lane)
    {
        constexpr int NCNT = FRG_NUM_XCD * (FRG_BWD_LEN_BUCKETS + 1);
        for (int i = lane; i < NCNT; i += 64) s->cnt[i] = bwd_cnt[i];
        __syncthreads();
        if (lane < FRG_NUM_XCD) {
            uint32_t sum = 0;
#pragma unroll 1
            for (int k = 0; k <= FRG_BWD_LEN_BUCKETS; k++) sum += s->cnt[lane * (FRG_BWD_LEN_BUCKETS + 1) + k];
            s->own[lane] = sum;
        }
        __syncthreads();
        if (lane == 0) {
            uint32_t N = 0;
            for (int x = 0; x < FRG_NUM_XCD; x++) N += s->own[x];
            // even shares: m_x = N / 8 (+ 1 for the first N % 8); donors' excess and takers' deficits line up in one pool
            uint32_t taken = 0, given = 0;
            for (int x = 0; x < FRG_NUM_XCD; x++) {
                const uint32_t m = N / FRG_NUM_XCD + ((uint32_t)x < N % FRG_NUM_XCD ? 1u : 0u), own = s->own[x];
                s->m[x] = m;
                if (own < m) { s->pool[x] = taken; taken += m - own; } else { s->pool[x] = given; given += own - m; }
            }
        }
        __syncthreads();
    }
};

// 4-bit version for the tile-per-wave backward: bit q <=> quadrant q may be touched
__device__ __forceinline__ uint32_t quadrant_mask(float x, float y, float4 co, int tx, int ty)
{
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (quadrant_hit(x, y, co, tx * FRG_TILE + (q & 1) * 8, ty * FRG_TILE + (q >> 1) * 8)) m |= 1u << q;
    return m;
}

// lane -> pixel of quadrant q inside the tile (backward: 4 pixels per lane)
__device__ __forceinline__ void lane_pixel(int lane, int q, int tx, int ty, int& px, int& py)
{
    px = tx * FRG_TILE + (q & 1) * 8 + (lane & 7);
    py = ty * FRG_TILE + (q >> 1) * 8 + (lane >> 3);
}

// ---------------------------------------------------------------------------
// one step of a fixed DPP reduction tree (v + the lane selected by CTRL)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v)
{
    // old = 0 with bound_ctrl: lets the compiler fold the DPP move into v_add_f32_dpp
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}

// a, b: two per-lane values -> one register whose low half holds a[l] + a[l+32] and whose high
// half holds b[l-32] + b[l] (v_permlane32_swap_b32: the upper 32 lanes of the first operand are
// exchanged with the lower 32 lanes of the second)
__device__ __forceinline__ float fold_two(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// BWD_BATCH: instances whose partial sums are reduced together (2 or 3: 18 / 27 matrix rows, two lanes per row)
// One wave per workgroup; the waves of XCD x take that XCD's (tile, segment) items (listed by the forward blend: the full
// segments first, then the tiles' last segments by decreasing length) by a static rule -- see the item loop.  An item walks
// the list positions [seg * SEG, min((seg + 1) * SEG, walked)) back to front.
template <bool EXACT, int BWD_BATCH>
__global__ void __launch_bounds__(64, EXACT ? 3 : 4)      // default arithmetic: 128 VGPRs without a spill instead of 130 -- the 16th wave per CU
blend_bwd_kernel(int T, int gx, int gy, int W, int H, const uint2* __restrict__ ranges,
                 const uint32_t* __restrict__ point_list, const float4* __restrict__ xydr,
                 const float4* __restrict__ conic_opacity, const float4* rgb_clamped /* one byte of .w is written: no restrict */,
                 const uint32_t* __restrict__ point_offsets, const float* __restrict__ bg,
                 const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                 const float* __restrict__ dL_dpix, float* __restrict__ slots, uint2* __restrict__ cutoff,
                 const uint32_t* __restrict__ bwd_cnt, const uint32_t* __restrict__ bwd_last, uint32_t cap_b,
                 const uint32_t* __restrict__ tile_work,
                 const char* __restrict__ binning_base, const Counters* __restrict__ counters, const float4* __restrict__ final_C,
                 int as_stamped)
{
    using M = BlendMath<EXACT>;
    const int lane = threadIdx.x;
    // What the forward's blend kernel stamped into the image chunk: a forward that kept nothing for a backward leaves no work
    // (and no valid checkpoints); and when the host does not state the arithmetic (as_stamped: it launches BOTH instantiations,
    // api.hip) only the one whose arithmetic is the forward's runs -- the stamp travels with the buffers, wherever they were
    // copied to and however long ago the forward ran.
    {
        const uint32_t flags = counters->fwd_flags;
        if (flags & FRG_FWD_ONLY) return;
        if (as_stamped && ((flags & FRG_FWD_EXACT) != 0u) != EXACT) return;
    }
    // The forward's checkpoints and its list of full-segment items: behind point_list and pairs of the chunk as the FORWARD
    // carved it (Counters::carved_R) -- the R this backward was called with may be the frame's instance count or a deferred
    // forward's capacity; the segment length is the one the forward blend stamped.
    const uint32_t carved_R = counters->carved_R;
    const int seg_log = (int)counters->bwd_seg_log;
    const float4* __restrict__ ckpt = reinterpret_cast<const float4*>(binning_base + BinningState::ckpt_offset(carved_R));
    const uint2* __restrict__ bwd_full = reinterpret_cast<const uint2*>(binning_base + BinningState::full_offset(carved_R));
    const uint32_t cap_a = (uint32_t)BinningState::full_cap(carved_R > 0u ? carved_R : 1u);

    __shared__ float4 s_a[64];     // x, y, quadrant mask, 0-based list position
    __shared__ float4 s_co[64];
    __shared__ float4 s_rgb[64];   // r, g, b, Gaussian-major slot index
    // Reduction matrix, one column per lane.  (Measured in round 3, both slower: rows padded to 68 floats make the
    // eight ds_read_b128 conflict-free -- 32 instead of 56 LDS cycles per batch in the bank model of
    // MI355X_MICROARCH.md -- but 16 workgroups x 10 416 B no longer fit the CU's 160 KiB and the 16th wave per CU
    // is worth more than the conflicts: 0.389 -> 0.407 ms; an XOR swizzle of the chunks by the row number keeps
    // the size but needs eight address registers: 130 VGPRs, three waves per SIMD.)
    __shared__ __attribute__((aligned(16))) float s_red[BWD_BATCH * FRG_SLOT_FLOATS * 64];   // reduction matrix, one column per lane
    // the XCDs' shares of the frame's items (BwdShares) live in the reduction matrix's space, which no item uses before
    // its first batch: built at the head of every round of the item loop (one round for nearly every wave)
    static_assert(sizeof(BwdShares) <= sizeof(float) * BWD_BATCH * FRG_SLOT_FLOATS * 64, "BwdShares must fit the reduction matrix");
    BwdShares* sh = reinterpret_cast<BwdShares*>(s_red);
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const size_t plane = (size_t)H * W;

    // ---- the item loop: static, no queue (see BwdShares) ----
    // Wave w of the W waves of XCD x takes the items k = r W + w (r even) / r W + (W - 1 - w) (r odd) of its XCD's m
    // items, r = 0, 1, ...: the items are in decreasing length, so the boustrophedon gives the wave with the longest
    // item of one round the shortest of the next.  The grid is usually larger than the frame has items (every wave then
    // has at most one: the hardware's dispatcher does the balancing, as it did with one workgroup per tile).
    const int my_xcd = (int)(blockIdx.x % FRG_NUM_XCD);
    const uint32_t W_ = gridDim.x / FRG_NUM_XCD, w_ = blockIdx.x / FRG_NUM_XCD;
    uint32_t my_m = 1u;               // (known after the first build)
  for (uint32_t round = 0; round * W_ < my_m; round++) {
    __syncthreads();               // the previous item's readers of the reduction matrix and the staging arrays are done
    BwdShares::build(sh, bwd_cnt, lane);
    my_m = sh->m[my_xcd];
    const uint32_t my_own = sh->own[my_xcd], my_pool = sh->pool[my_xcd];
    const uint32_t k = round * W_ + ((round & 1u) ? W_ - 1u - w_ : w_);
    // item k of this XCD: one of its own, or -- an XCD with fewer than its share -- one from the pool of the others' excess
    int src = k < my_m ? my_xcd : -1;  // (k >= my_m: the last, partial round)
    uint32_t ks = k;
    if (src >= 0 && k >= my_own) {
        const uint32_t j = my_pool + (k - my_own);
        src = -1;
        for (int d = 0; d < FRG_NUM_XCD; d++) {
            const uint32_t own = sh->own[d], m = sh->m[d], first = sh->pool[d];
            if (own > m && j >= first && j < first + (own - m)) { src = d; ks = m + (j - first); }
        }
    }
    int tile = -1;
    uint32_t seg = 0xFFFFFFFFu;        // 0xFFFFFFFF: the tile's last segment
    if (src >= 0) {
        const uint32_t* c = sh->cnt + src * (FRG_BWD_LEN_BUCKETS + 1);
        const uint32_t na = c[FRG_BWD_LEN_BUCKETS];             // (<= cap_a: the forward sized the list for R / SEG items, and clamps)
        if (ks < na) { if (ks < cap_a) { const uint2 it = bwd_full[(size_t)src * cap_a + ks]; tile = (int)it.x; seg = it.y; } }
        else {
            uint32_t j = ks - na;
#pragma unroll 1
            for (int bk = 0; bk < FRG_BWD_LEN_BUCKETS; bk++) {
                const uint32_t cb = c[bk];
                if (j < cb) { if (j < cap_b) tile = (int)bwd_last[((size_t)src * FRG_BWD_LEN_BUCKETS + bk) * cap_b + j]; break; }
                j -= cb;
            }
        }
    }
    __syncthreads();                   // the shares are dead: the reduction matrix is free
    if (tile < 0) continue;            // wave-uniform (the last, partial round; cannot happen otherwise: the pool is exactly the excess)
    const uint32_t walked = tile_work[tile];
    if (seg == 0xFFFFFFFFu) seg = (walked - 1u) >> seg_log;
    const int tx = tile % gx, ty = tile / gx;
    const uint2 rg = ranges[tile];
    const int seg_lo = (int)(seg << seg_log);
    const uint32_t seg_end = min(walked, (seg + 1u) << seg_log);        // exclusive; == walked for the tile's last segment
    const bool last_seg = seg_end == walked;

    // Per-pixel state of the back-to-front walk.  The reference carries the colour
    // composited behind the current Gaussian (accum_rec, last_color, last_alpha:
    // backward.cu:508-521); we carry the equivalent scalar
    //     S = sum_ch dL/dC_ch * (colour already composited behind) + T_final * (bg . dL/dC)
    // so that dL/dalpha_i = T_i * (c_i . dL/dC) - S / (1 - alpha_i), then S += alpha_i T_i (c_i . dL/dC).
    // Same mathematics, 2 registers instead of 9 per pixel.
    // A pixel whose last contributor lies behind this segment starts from the forward's checkpoint at the segment's
    // end: T = the transmittance there, colour behind = (colour the pixel ended with) - (colour accumulated there).
    float pxf[4], pyf[4], Tr[4], S[4], dLp[4][3];
    uint32_t lastcon[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int px, py;
        lane_pixel(lane, q, tx, ty, px, py);
        pxf[q] = (float)px; pyf[q] = (float)py;
        const bool inside = px < W && py < H;
        const size_t pid = (size_t)py * W + px;
        Tr[q] = inside ? final_T[pid] : 0.0f;
        lastcon[q] = inside ? n_contrib[pid] : 0u;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) dLp[q][ch] = inside ? dL_dpix[ch * plane + pid] : 0.0f;
        S[q] = Tr[q] * M::mad(bg2, dLp[q][2], M::mad(bg1, dLp[q][1], bg0 * dLp[q][0]));
    }
    if (!last_seg) {               // wave-uniform: some pixel may go on behind this segment
        const float4* ck = ckpt + ((size_t)(rg.x >> seg_log) + (size_t)(seg + 1u)) * FRG_TILE_PIX;
        const float4* fc = final_C + (size_t)tile * FRG_TILE_PIX;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (lastcon[q] > seg_end) {
                const float4 c = ck[q * 64 + lane], f = fc[q * 64 + lane];
                const float behind = M::mad(f.z - c.w, dLp[q][2], M::mad(f.y - c.z, dLp[q][1], (f.x - c.y) * dLp[q][0]));
                S[q] = S[q] + behind;          // (S held T_final (bg . dL/dC) so far)
                Tr[q] = c.x;
            }
        }
    }
    // per-quadrant number of list entries that can still receive gradient, cut at the segment's end
    uint32_t qmax[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        qmax[q] = lastcon[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) qmax[q] = max(qmax[q], (uint32_t)__shfl_xor((int)qmax[q], d, 64));
        qmax[q] = min(qmax[q], seg_end);
    }
    const uint32_t maxc = seg_end;     // (== max over the quadrants for the last segment: walked IS that maximum)
    if (last_seg && lane == 0) {
        const uint32_t id = point_list[rg.x + walked - 1];
        cutoff[tile] = make_uint2(__float_as_uint(xydr[FRG_REC * id].z), id);
    }

    // walk the segment [seg_lo, maxc) back to front, 64 instances at a time.  The walk is known in advance (no
    // early termination), so the records of round r + 1 -- id, three 16-byte gathers, the instance offset: a ~4 us
    // dependent chain -- are requested before round r is processed.  The loads are unconditional (clamped
    // positions): a conditional load into a loop-carried register makes the compiler copy it, and wait, at once.
    uint32_t id_n = 0, off_n = 0;
    float4 a_n, co_n, col_n;
    auto fetch = [&](int hi) {
        id_n = point_list[rg.x + max(hi - lane, seg_lo)];
        a_n = xydr[FRG_REC * id_n];
        co_n = conic_opacity[FRG_REC * id_n];
        col_n = rgb_clamped[FRG_REC * id_n];
        off_n = point_offsets[max(id_n, 1u) - 1u];
    };
    fetch((int)maxc - 1);
    for (int hi = (int)maxc - 1; hi >= seg_lo; hi -= 64) {
        const int cnt = min(64, hi - seg_lo + 1);
        uint32_t m = 0, my_slot = 0;
        const uint32_t id = id_n;
        const float4 a = a_n, co = co_n, col = col_n;
        const uint32_t off = id == 0 ? 0u : off_n;
        if (hi - 64 >= seg_lo) fetch(hi - 64);    // wave-uniform
        if (lane < cnt) {
            const uint32_t mypos = (uint32_t)(hi - lane);
            m = quadrant_mask(a.x, a.y, co, tx, ty) &
                ((mypos < qmax[0] ? 1u : 0u) | (mypos < qmax[1] ? 2u : 0u) | (mypos < qmax[2] ? 4u : 0u) | (mypos < qmax[3] ? 8u : 0u));
            // Gaussian-major slot of this (Gaussian, tile) instance: position in the
            // reference's duplicateWithKeys emission order (rasterizer_impl.cu:98-108)
            int x0, y0, x1, y1;
            tile_rect(a.x, a.y, (int)a.w, gx, gy, x0, y0, x1, y1);
            my_slot = off + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
            if (m == 0) {  // provably no contribution in this tile: the slot is still owed a value
                float* dst = slots + (size_t)my_slot * FRG_SLOT_STRIDE;
#pragma unroll
                for (int c = 0; c < FRG_SLOT_FLOATS; c++) dst[c] = 0.0f;
            } else if (!(__float_as_uint(col.w) & FRG_REACHED_MASK)) {
                // "this Gaussian has a slot that may hold a gradient": byte 1 of the record's flag word (the forward writes
                // the word, clamp flags in byte 0, with every record).  The per-Gaussian backward reduces the slots of the
                // marked Gaussians only.  Every writer stores the same value, and the set of marks is a function of the
                // forward state alone (the cull bound and the pixels' last contributors, neither depends on the blend
                // arithmetic or on dL/dpixel): a backward repeated on the same state finds exactly the marks it would set.
                reinterpret_cast<uint8_t*>(const_cast<float4*>(rgb_clamped + FRG_REC * id))[13] = 1;
            }
        }
        const uint64_t keep = wave_ballot(m != 0);
        const int nkeep = __popcll(keep);
        __syncthreads();
        if (m != 0) {
            const int d = lanes_before(keep, lane);
            s_a[d] = make_float4(a.x, a.y, __uint_as_float(m), __uint_as_float((uint32_t)(hi - lane)));
            s_co[d] = M::stage(co);
            s_rgb[d] = make_float4(col.x, col.y, col.z, __uint_as_float(my_slot));
        }
        __syncthreads();
        // BWD_BATCH surviving instances per iteration; their BWD_BATCH x 9 per-lane partial sums are
        // reduced across the wave THROUGH LDS: every lane stores its 27 partials as one column of a
        // [27][64] matrix (conflict-free ds_write), then two lanes per row add up 32 entries each
        // (8 ds_read_b128, chunk order skewed by the row so that the rows spread over the banks) and one
        // DPP add joins the halves.  Measured on gfx950 (tools/micro/pk_rate.hip): v_add_f32 2.8 cycles per
        // wave-instruction, v_add_f32_dpp 7, v_permlane32_swap 12.6 -- the former swap + DPP tree cost
        // ~240 cycles per instance, almost as much as the per-pixel mathematics; this form ~75.
        for (int k = 0; k < nkeep; k += BWD_BATCH) {
            float acc[BWD_BATCH][FRG_SLOT_FLOATS];
#pragma unroll
            for (int h = 0; h < BWD_BATCH; h++)
#pragma unroll
                for (int c = 0; c < FRG_SLOT_FLOATS; c++) {
                    acc[h][c] = 0.0f;
                    // opaque to the optimiser: knowing the zero, it keeps one set of nine temporaries per quadrant
                    // (zeroed again on every skipped quadrant) and adds them up afterwards
                    asm volatile("" : "+v"(acc[h][c]));
                }
            bool any = false;
#pragma unroll
            for (int h = 0; h < BWD_BATCH; h++) {
                const int kk = k + h;
                if (kk >= nkeep) break;
                float* part = acc[h];
                const float4 ca = s_a[kk], cco = s_co[kk];
                const uint32_t qm = __builtin_amdgcn_readfirstlane(__float_as_uint(ca.z));
                const uint32_t pos = __float_as_uint(ca.w);  // 0-based position in the tile list
                const float4 gc = s_rgb[kk];
                // per quadrant the cull kept (wave-uniform mask): falloff, the reference's three tests, and --
                // only if some pixel of the quadrant blended this Gaussian -- its gradient contributions
                // (part[3..8] are pixel MOMENTS of v = G dL/dalpha; opacity, conic and the NDC factors are applied
                // once per Gaussian by the per-Gaussian backward, after it has summed the instances)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (!(qm & (1u << q))) continue;
                    float dx, dy;
                    const float power = M::power(ca.x, ca.y, cco, pxf[q], pyf[q], dx, dy);
                    const float G = M::expo(power);
                    const float alpha = fminf(0.99f, cco.w * G);
                    const bool ok = pos < lastcon[q] && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                    if (wave_ballot(ok) == 0ull) continue;       // wave-uniform
                    any = true;
                    // No per-lane branch below: a pixel that did not blend this Gaussian runs the same
                    // instructions with alpha = G = 0, which leaves its state and every sum exactly unchanged
                    // (T * 1, S + 0 * x, sums + 0).  A divergent `if (ok)` made the compiler merge nine zero-
                    // initialised temporaries per quadrant into the accumulators (18 extra instructions of ~60).
                    const float a_eff = ok ? alpha : 0.0f, g_eff = ok ? G : 0.0f;
                    const float rinv = M::recip(1.f - a_eff);    // 1 - alpha >= 0.01; exactly 1 for a_eff = 0
                    Tr[q] = Tr[q] * rinv;                        // transmittance in front of this Gaussian
                    const float w = a_eff * Tr[q];               // dC/dcolour
                    const float cdot = M::mad(gc.z, dLp[q][2], M::mad(gc.y, dLp[q][1], gc.x * dLp[q][0]));
                    part[0] = M::mad(w, dLp[q][0], part[0]);
                    part[1] = M::mad(w, dLp[q][1], part[1]);
                    part[2] = M::mad(w, dLp[q][2], part[2]);
                    const float dL_dalpha = M::mad(Tr[q], cdot, -(S[q] * rinv));
                    S[q] = M::mad(w, cdot, S[q]);
                    const float v = g_eff * dL_dalpha;
                    // moments about the Gaussian's centre, the reference's d (backward.cu:441,536-554), in both arithmetics.
                    // (Round 3's default arithmetic took them about the TILE centre -- one FMA each on per-lane constants, 6
                    // instead of 8 instructions -- and shifted every slot to its Gaussian afterwards: 5 us faster at C3, and
                    // up to 5 x farther from the float64 gradient on sparse frames, profiles/r04_sparse_grad_check_with_r03_forms.log.)
                    const float vx = v * dx, vy = v * dy;
                    part[3] += vx;
                    part[4] += vy;
                    part[5] = M::mad(vx, dx, part[5]);
                    part[6] = M::mad(vx, dy, part[6]);
                    part[7] = M::mad(vy, dy, part[7]);
                    part[8] += v;
                }
            }
            const int row = lane >> 1, half = lane & 1;
            const int inst = row / FRG_SLOT_FLOATS;              // which instance of the batch
            const int comp = row - inst * FRG_SLOT_FLOATS;
            // the lane that ends up with (instance, component) stores it straight into the instance's slot:
            // the nine lanes of an instance write 36 consecutive bytes
            const bool writer = half == 0 && row < BWD_BATCH * FRG_SLOT_FLOATS && k + inst < nkeep;
            float* dst = writer ? slots + (size_t)__float_as_uint(s_rgb[k + inst].w) * FRG_SLOT_STRIDE + comp : nullptr;
            if (!any) {              // nothing blended in this batch: its slots are still owed their zeros
                if (writer) *dst = 0.0f;
                continue;
            }
            wave_lds_sync();         // the previous batch's readers are done with s_red
#pragma unroll
            for (int h = 0; h < BWD_BATCH; h++)
#pragma unroll
                for (int c = 0; c < FRG_SLOT_FLOATS; c++) s_red[(h * FRG_SLOT_FLOATS + c) * 64 + lane] = acc[h][c];
            wave_lds_sync();
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if (row < BWD_BATCH * FRG_SLOT_FLOATS) {
                const float4* src = reinterpret_cast<const float4*>(s_red + row * 64 + half * 32);
                // chunk order skewed by the COMPONENT, not by the row: the order in which an instance's partials
                // are added must not depend on its place in the batch (hence in the tile list)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float4 v = src[(j + comp) & 7];
                    s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
                }
            }
            float sum = (s0 + s1) + (s2 + s3);
            sum = dpp_step<0xB1, 0xf>(sum);                      // quad_perm [1,0,3,2]: the row's other half
            if (writer) *dst = sum;
        }
    }
  }   // next item
}


// ---- launchers (instantiated by blend_exact.hip / blend_fast.hip with their arithmetic) ----------------------------------
extern int g_fwd_order;       // tuning (frg_set_option("fwd_order")): 1 = forward blend walks the tiles longest list first
template <bool EXACT>
static hipError_t launch_blend_fwd_t(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                     const float* bg, float* out_color, bool prefetch, hipStream_t s, bool forward_only = false,
                                     bool fused_sort = false, bool long_lists = false)
{
    const int T = vp.gx * vp.gy;
#define FRG_FWD(PF, FS, UN)                                                                                                \
    hipLaunchKernelGGL((blend_fwd_kernel<EXACT, PF, FS, UN>), dim3(xcd_grid_blocks(T)), dim3(BLEND_THREADS), 0, s, T, vp.gx, vp.W, vp.H, \
                       img.ranges, b.point_list, g.xydr, g.conic_opacity, g.rgb_clamped, bg, img.final_T, img.n_contrib,   \
                       out_color, img.tile_work, forward_only ? nullptr : b.ckpt, img.final_C, g_fwd_order ? img.class_tiles : nullptr,             \
                       img.counters->class_count, b.seg_log, img.bwd_cnt, img.bwd_last, img.bwd_cap_b, b.bwd_full,         \
                       (uint32_t)BinningState::full_cap(b.carved_R), img.cutoff, img.counters, b.pairs)
    // long_lists (the host's reading of the frame's counters): the work sits in a few long lists -> eight entries per trip
    if (fused_sort) { if (prefetch) FRG_FWD(true, true, 4); else FRG_FWD(false, true, 4); }
    else if (long_lists && prefetch) FRG_FWD(true, false, 8);
    else if (prefetch) FRG_FWD(true, false, 4); else FRG_FWD(false, false, 4);
#undef FRG_FWD
    return hipGetLastError();
}

// R: the instance count the caller sized the slots for -- an upper bound on the frame's items: one last segment per tile +
// R / (shortest segment) full ones; waves: the single-wave workgroups of the segmented form (16 per CU fit the LDS)
// measured, backward blend at C3 / C4: segments of 1024 -- 2048 waves 0.69 / 0.56 ms, 4096 (= what is resident at once) with a
// queue 0.55 / 0.39, 8192 0.41 / 0.39; segments of 512 (13 000 items at C3) -- 8192 waves 0.386 / 0.370, 16384 0.363 / 0.368,
// 32768 0.366 / 0.371: one item per wave and the hardware's dispatcher, while the items fit the grid
#define FRG_BWD_MAX_WAVES 16384
extern int g_bwd_waves;       // tuning (frg_set_option("bwd_waves")): single-wave workgroups of the backward blend (0: the default)
template <bool EXACT>
static hipError_t launch_blend_bwd_t(const ViewParams& vp, const GeomState& g, const ImageState& img, const BinningState& b,
                                     const float* bg, const float* dL_dpix, float* slots, uint32_t R, int batch, hipStream_t s, bool as_stamped)
{
    const int T = vp.gx * vp.gy;
    // waves: one per item while the frame has at most FRG_BWD_MAX_WAVES items (an upper bound on their number), beyond
    // that the waves stride
    const int bound = (int)std::min<size_t>((size_t)T + BinningState::full_cap(R), (size_t)FRG_BWD_MAX_WAVES);
    const int nwaves = ((g_bwd_waves > 0 ? g_bwd_waves : bound) + 7) / 8 * 8;
#define FRG_BWD(B)                                                                                                         \
    hipLaunchKernelGGL((blend_bwd_kernel<EXACT, B>), dim3(nwaves), dim3(64), 0, s, T, vp.gx, vp.gy, vp.W, vp.H,               \
                       img.ranges, b.point_list, g.xydr, g.conic_opacity, g.rgb_clamped, g.point_offsets, bg, img.final_T,     \
                       img.n_contrib, dL_dpix, slots, img.cutoff, img.bwd_cnt, img.bwd_last, img.bwd_cap_b, img.tile_work,     \
                       reinterpret_cast<const char*>(b.point_list), img.counters, img.final_C, as_stamped ? 1 : 0)
    if (batch == 2) FRG_BWD(2); else FRG_BWD(3);
#undef FRG_BWD
    return hipGetLastError();
}

}  // namespace frg
