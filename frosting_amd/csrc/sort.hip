// Per-tile depth sort in LDS.
//
// Replaces cub::DeviceRadixSort::SortPairs over all R (tile|depth) keys
// (rasterizer_impl.cu:303-308, ~6 global radix passes of 24 B/instance) with one
// workgroup per tile that loads its segment (8 B/instance), sorts it with a
// stable wave64 ballot-ranked LSD radix sort on the 32 depth bits, breaks depth
// ties by ascending Gaussian index and writes the 4 B/instance point_list.
// Result == the reference's stable sort of index-ordered (tile,depth) keys.
#include "frg_common.h"
#include "kernels.h"

#include <atomic>

namespace frg {

// Lanes of the wave holding the same 8-bit digit as this lane (invalid lanes excluded).  Per bit: the lane's bit
// sign-extended to a mask sb (v_bfe_i32), one ballot, and peers &= ~(ballot ^ sb) per 32-bit half (v_xnor + v_and):
// six vector instructions.  (`peers &= bit ? m : ~m` on 64-bit values compiled to nine: the ranking is the sort's
// instruction-bound inner loop.)
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid)
{
    const uint64_t v = __builtin_amdgcn_ballot_w64(valid);
    uint32_t plo = (uint32_t)v, phi = (uint32_t)(v >> 32);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t sb = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);   // 0 or 0xFFFFFFFF
        const uint64_t m = __builtin_amdgcn_ballot_w64(sb != 0u);
        plo &= ~((uint32_t)m ^ sb);
        phi &= ~((uint32_t)(m >> 32) ^ sb);
    }
    return ((uint64_t)phi << 32) | plo;
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask, int lane)
{
    return (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Depth ties: order equal-depth runs by ascending index (the stable-sort tie rule,
// SURVEY Appendix A-7).  count_ties() returns the number of adjacent equal-depth
// pairs; a handful are fixed by insertion (fix_ties), many (coplanar scenes) by
// re-sorting on the index first and the depth again, which LSD stability turns
// into (depth, index) order.
template <typename PtrT>
__device__ __forceinline__ int count_ties(PtrT a, int n, int nthreads, uint32_t* scratch)
{
    if (threadIdx.x == 0) scratch[260] = 0;
    __syncthreads();
    uint32_t c = 0;
    for (int i = threadIdx.x; i < n - 1; i += nthreads) c += (a[i].x == a[i + 1].x) ? 1u : 0u;
    if (c) atomicAdd(&scratch[260], c);
    __syncthreads();
    const int r = (int)scratch[260];
    __syncthreads();
    return r;
}

template <typename PtrT>
__device__ __forceinline__ void fix_ties(PtrT a, int n, int nthreads)
{
    for (int i = threadIdx.x; i < n - 1; i += nthreads) {
        const uint32_t k = a[i].x;
        if (a[i + 1].x != k) continue;
        if (i > 0 && a[i - 1].x == k) continue;  // not the run start
        int j = i + 1;
        while (j + 1 < n && a[j + 1].x == k) j++;
        for (int p = i + 1; p <= j; p++) {  // insertion sort on .y within [i, j]
            const uint2 v = a[p];
            int q = p - 1;
            while (q >= i && a[q].y > v.y) { a[q + 1] = a[q]; q--; }
            a[q + 1] = v;
        }
    }
    __syncthreads();
}

// ---- LDS classes: register-staged, in-place passes --------------------------------
// Each thread owns CAP / threads (8 or 16) elements of its wave's contiguous strip.
// A pass ranks them from REGISTERS (no LDS reads of the data), scatters them into the
// single LDS buffer and reloads its strip: one 8-byte buffer instead of a ping-pong pair,
// so twice the workgroups fit per CU and different size classes can share a CU.
template <int NWAVES, int SORT_ITEMS, bool BY_INDEX>
__device__ __forceinline__ bool radix_pass_regs(uint2 (&e)[SORT_ITEMS], int n, int begin, int end, int shift,
                                                uint2* buf, uint32_t* whist, uint32_t* scratch)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* myhist = whist + wave * 256;
#pragma unroll
    for (int i = 0; i < 4; i++) myhist[lane * 4 + i] = 0;   // wave-private: no workgroup barrier needed
    uint32_t meta[SORT_ITEMS];  // rank within the 64-element step | group size << 8 | digit << 16
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; it++) {
        meta[it] = 0;
        if (begin + it * 64 >= end) continue;     // wave-uniform: nothing of the strip in this step
        const bool valid = begin + it * 64 + lane < end;
        const uint32_t d = ((BY_INDEX ? e[it].y : e[it].x) >> shift) & 255u;
        const uint64_t peers = match_digit(d, valid);
        const uint32_t rank = lanes_below(peers, lane), cnt = (uint32_t)__popcll(peers);
        meta[it] = rank | (cnt << 8) | (d << 16);
        if (valid && rank == 0) atomicAdd(&myhist[d], cnt);    // ds_add_u32: no read / wait / write-back round trip
    }
    __syncthreads();
    // digit totals, exclusive over waves then over digits
    constexpr int NT = NWAVES * 64;
    for (int dg = tid; dg < 256; dg += NT) {      // (a 64-thread workgroup covers the 256 digits in 4 steps)
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; w++) {
            const uint32_t c = whist[w * 256 + dg];
            whist[w * 256 + dg] = run;
            run += c;
        }
        scratch[dg] = run;
        // one digit holds everything: nothing to move.  One flag per pass position (reset once per group of
        // passes by the caller), so no barrier is spent on clearing it
        if (run == (uint32_t)n) scratch[256 + (shift >> 3)] = 1;
    }
    __syncthreads();   // (every thread also holds its elements in registers by now: the buffer may be overwritten)
    if (scratch[256 + (shift >> 3)] != 0) return false;
    {   // EVERY wave scans the 256 digit totals itself (4 per lane, DPP) and adds the digit bases to its own cursor
        // row: no single-wave scan with a barrier on either side, three workgroup barriers per pass instead of six
        const uint32_t v0 = scratch[4 * lane], v1 = scratch[4 * lane + 1], v2 = scratch[4 * lane + 2], v3 = scratch[4 * lane + 3];
        const uint32_t s4 = v0 + v1 + v2 + v3;
        const uint32_t ex = wave_incl_scan_dpp(s4) - s4;
        myhist[4 * lane] += ex; myhist[4 * lane + 1] += ex + v0;
        myhist[4 * lane + 2] += ex + v0 + v1; myhist[4 * lane + 3] += ex + v0 + v1 + v2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; it++) {
        if (begin + it * 64 >= end) break;        // wave-uniform
        const bool valid = begin + it * 64 + lane < end;
        const uint32_t rank = meta[it] & 255u, cnt = (meta[it] >> 8) & 255u, d = meta[it] >> 16;
        uint32_t pos = 0;
        if (valid) pos = myhist[d] + rank;
        __builtin_amdgcn_wave_barrier();          // all lanes read the cursor before a leader bumps it
        if (valid && rank == 0) atomicAdd(&myhist[d], cnt);
        if (valid) buf[pos] = e[it];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; it++) {
        const int i = begin + it * 64 + lane;
        if (i < end) e[it] = buf[i];
    }
    return true;
}

// CAP = 8 * threads; tiles with lo < n <= CAP are handled by this instantiation (the host
// launches one instantiation per size class; workgroups whose tile is outside exit at once).
template <int NWAVES, int CAP>
__global__ void __launch_bounds__(NWAVES * 64)
sort_tiles_lds_kernel(const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ list_len,
                      const uint2* __restrict__ ranges, const uint2* __restrict__ pairs, uint32_t* __restrict__ point_list)
{
    constexpr int SORT_ITEMS = CAP / (NWAVES * 64);   // elements staged per thread
    static_assert(CAP == NWAVES * 64 * SORT_ITEMS && SORT_ITEMS >= 1, "CAP must be a multiple of the workgroup size");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2* buf = reinterpret_cast<uint2*>(smem);
    uint32_t* whist = reinterpret_cast<uint32_t*>(buf + CAP);
    uint32_t* scratch = whist + NWAVES * 256;  // 264 words: 256 digit totals, 4 uniform-pass flags, tie counter
    constexpr int NT = NWAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the grid normally covers exactly the tiles of this size class (one pass of this loop); when the
    // host only had an estimate of the list length, workgroups stride over the device-side list
    const uint32_t ntiles = *list_len;
    for (uint32_t item = blockIdx.x; item < ntiles; item += gridDim.x) {
        if (item != blockIdx.x) __syncthreads();   // the previous tile's LDS contents are dead
        const int tile = (int)tile_list[item];
        const uint2 rg = ranges[tile];
        const int n = (int)(rg.y - rg.x);
        if (n > CAP) continue;
        // contiguous strip per wave (keeps every pass stable), multiple of 64, at most 64 * SORT_ITEMS
        const int strip = ((n + NWAVES - 1) / NWAVES + 63) & ~63;
        const int begin = wave * strip, end = min(n, begin + strip);
        uint2 e[SORT_ITEMS];
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; it++) {
            const int i = begin + it * 64 + lane;
            e[it] = i < end ? pairs[rg.x + i] : make_uint2(0u, 0u);
        }
        if (tid < 4) scratch[256 + tid] = 0;
        __syncthreads();
        bool in_lds = false;
        if (n > 1) {
#pragma unroll 1
            for (int pass = 0; pass < 4; pass++)
                in_lds |= radix_pass_regs<NWAVES, SORT_ITEMS, false>(e, n, begin, end, 8 * pass, buf, whist, scratch);
        }
        if (!in_lds) {   // nothing moved (n == 1 or all keys equal): materialise the strip for the steps below
#pragma unroll
            for (int it = 0; it < SORT_ITEMS; it++) {
                const int i = begin + it * 64 + lane;
                if (i < end) buf[i] = e[it];
            }
            __syncthreads();
        }
        if (n > 1) {
            const int ties = count_ties<uint2*>(buf, n, NT, scratch);
            if (ties > 32) {
                // many equal depths (coplanar scenes): order by index, then by depth again -- LSD
                // stability turns that into (depth, index)
                if (tid < 4) scratch[256 + tid] = 0;      // (count_ties ended with a barrier; the first pass has one before the flags are used)
#pragma unroll 1
                for (int pass = 0; pass < 4; pass++) radix_pass_regs<NWAVES, SORT_ITEMS, true>(e, n, begin, end, 8 * pass, buf, whist, scratch);
                if (tid < 4) scratch[256 + tid] = 0;      // (every thread is past the last pass's flag; the next read is two barriers away)
#pragma unroll 1
                for (int pass = 0; pass < 4; pass++) radix_pass_regs<NWAVES, SORT_ITEMS, false>(e, n, begin, end, 8 * pass, buf, whist, scratch);
            } else if (ties > 0) {
                fix_ties<uint2*>(buf, n, NT);
            }
        }
        for (int i = tid; i < n; i += NT) point_list[rg.x + i] = buf[i].y;
    }
}

// ---- lists longer than the LDS capacity: multi-workgroup LSD radix sort in global memory ----------------
// One workgroup per list takes milliseconds on a list of 10^5 entries -- clustered scenes
// have them.  Here every 1024-element strip of such a list is one WAVE's work: per 8-bit digit
//   big_count   each wave counts the digits of its strip -> one 256-entry row of big_hist
//   big_offsets per list: exclusive scan of the rows, digit-major then strip-major (stable)
//   big_move    each wave ranks its strip again (ballot matching keeps the strip's order) and scatters
// ping-ponging pairs <-> pairs_tmp.  Passes on the INDEX bits first, then on the 32 depth bits: LSD stability
// leaves equal depths in ascending index order, the reference's tie rule.  The last pass writes point_list.
#define BIG_STRIP 1024
#define BIG_THREADS 256    // 4 waves = 4 strips per workgroup

template <bool BY_INDEX>
__global__ void __launch_bounds__(BIG_THREADS)
big_count_kernel(const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ list_len,
                 const uint2* __restrict__ ranges, const uint2* __restrict__ src, int shift, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t whist[(BIG_THREADS / 64) * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* myhist = whist + wave * 256;
    // the grid normally covers every (list, strip) pair once; both loops stride so that a grid sized from an
    // estimate (deferred-counters forward) still covers everything
    for (uint32_t item = blockIdx.y; item < *list_len; item += gridDim.y) {
        const uint2 rg = ranges[tile_list[item]];
        const int n = (int)(rg.y - rg.x);
        for (int strip = blockIdx.x * (BIG_THREADS / 64) + wave; strip * BIG_STRIP < n; strip += gridDim.x * (BIG_THREADS / 64)) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; i++) myhist[lane * 4 + i] = 0;
            const int begin = strip * BIG_STRIP, end = min(n, begin + BIG_STRIP);
            for (int i = begin; i < end; i += 64) {
                const bool valid = i + lane < end;
                const uint2 e = valid ? src[rg.x + i + lane] : make_uint2(0u, 0u);
                const uint32_t d = ((BY_INDEX ? e.y : e.x) >> shift) & 255u;
                const uint64_t peers = match_digit(d, valid);
                if (valid && lanes_below(peers, lane) == 0) myhist[d] += (uint32_t)__popcll(peers);
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t* row = hist + BinningState::big_hist_row(rg.x, (uint32_t)strip) * 256;
#pragma unroll
            for (int i = 0; i < 4; i++) row[lane + 64 * i] = myhist[lane + 64 * i];
        }
    }
}

__global__ void __launch_bounds__(256)
big_offsets_kernel(const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ list_len,
                   const uint2* __restrict__ ranges, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t tot[256];
    const int d = threadIdx.x;
    for (uint32_t item = blockIdx.x; item < *list_len; item += gridDim.x) {
        const uint2 rg = ranges[tile_list[item]];
        const int n = (int)(rg.y - rg.x);
        const int nstrips = (n + BIG_STRIP - 1) / BIG_STRIP;
        uint32_t* rows = hist + BinningState::big_hist_row(rg.x, 0u) * 256;
        uint32_t run = 0;
        for (int r = 0; r < nstrips; r++) { const uint32_t c = rows[(size_t)r * 256 + d]; rows[(size_t)r * 256 + d] = run; run += c; }
        __syncthreads();
        tot[d] = run;
        __syncthreads();
        if (d < 64) {   // exclusive scan of the 256 digit totals by one wave, 4 per lane
            const uint32_t v0 = tot[4 * d], v1 = tot[4 * d + 1], v2 = tot[4 * d + 2], v3 = tot[4 * d + 3];
            const uint32_t s = v0 + v1 + v2 + v3;
            const uint32_t ex = wave_incl_scan_dpp(s) - s;
            tot[4 * d] = ex; tot[4 * d + 1] = ex + v0; tot[4 * d + 2] = ex + v0 + v1; tot[4 * d + 3] = ex + v0 + v1 + v2;
        }
        __syncthreads();
        const uint32_t base = tot[d];
        for (int r = 0; r < nstrips; r++) rows[(size_t)r * 256 + d] += base;
    }
}

template <bool BY_INDEX, bool LAST>
__global__ void __launch_bounds__(BIG_THREADS)
big_move_kernel(const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ list_len,
                const uint2* __restrict__ ranges, const uint2* __restrict__ src, uint2* __restrict__ dst,
                uint32_t* __restrict__ point_list, int shift, const uint32_t* __restrict__ hist)
{
    __shared__ uint32_t whist[(BIG_THREADS / 64) * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* cursor = whist + wave * 256;
    for (uint32_t item = blockIdx.y; item < *list_len; item += gridDim.y) {
        const uint2 rg = ranges[tile_list[item]];
        const int n = (int)(rg.y - rg.x);
        for (int strip = blockIdx.x * (BIG_THREADS / 64) + wave; strip * BIG_STRIP < n; strip += gridDim.x * (BIG_THREADS / 64)) {
            const uint32_t* row = hist + BinningState::big_hist_row(rg.x, (uint32_t)strip) * 256;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; i++) cursor[lane + 64 * i] = row[lane + 64 * i];
            __builtin_amdgcn_wave_barrier();
            const int begin = strip * BIG_STRIP, end = min(n, begin + BIG_STRIP);
            for (int i = begin; i < end; i += 64) {
                const bool valid = i + lane < end;
                const uint2 e = valid ? src[rg.x + i + lane] : make_uint2(0u, 0u);
                const uint32_t d = ((BY_INDEX ? e.y : e.x) >> shift) & 255u;
                const uint64_t peers = match_digit(d, valid);
                const uint32_t rank = lanes_below(peers, lane);
                uint32_t pos = 0;
                if (valid) pos = cursor[d] + rank;
                __builtin_amdgcn_wave_barrier();                  // all lanes read the cursor before a leader bumps it
                if (valid && rank == 0) cursor[d] += (uint32_t)__popcll(peers);
                if (valid) {
                    if (LAST) point_list[rg.x + pos] = e.y;
                    else dst[rg.x + pos] = e;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// Host-side launcher (called from api.hip)
template <int NW, int CAP>
static hipError_t launch_lds_class(int count, const uint32_t* tile_list, const uint32_t* list_len, const uint2* ranges,
                                   const uint2* pairs, uint32_t* point_list, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    const size_t lds = (size_t)CAP * 8 + NW * 1024 + 264 * 4;
    if (lds > 48 * 1024) {
        // the attribute is per device: remember it per device, not per process
        static std::atomic<unsigned long long> attr_set{0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_set.load() & bit)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_tiles_lds_kernel<NW, CAP>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_set.fetch_or(bit);
        }
    }
    hipLaunchKernelGGL((sort_tiles_lds_kernel<NW, CAP>), dim3(count), dim3(NW * 64), lds, stream, tile_list, list_len, ranges, pairs, point_list);
    return hipGetLastError();
}

// The size classes touch disjoint tiles, so they run concurrently: the large-tile
// classes are forked onto two internal side streams (event fork / join around the
// caller's stream) instead of queueing behind the small-tile pass.  Each launch covers
// exactly the tiles of its class (lists built by scan_kernel): a grid of all T tiles with
// early exits was dominated by dispatching ~6000 no-op 1024-thread workgroups.
struct SortStreams {
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    int device = -1;
    bool ensure()
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (dev == device) return true;
        // first use on this thread, or the caller switched devices: (re)create on the current one
        for (int i = 0; i < 2; i++) {
            if (side[i]) (void)hipStreamDestroy(side[i]);
            if (join[i]) (void)hipEventDestroy(join[i]);
            side[i] = nullptr; join[i] = nullptr;
        }
        if (fork) (void)hipEventDestroy(fork);
        fork = nullptr;
        device = -1;
        for (int i = 0; i < 2; i++) {
            if (hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&join[i], hipEventDisableTiming) != hipSuccess) return false;
        }
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
        device = dev;
        return true;
    }
};

hipError_t launch_tile_sort(int T, const uint32_t* class_count, const uint32_t* grid_hint, const uint32_t* class_count_dev,
                            const uint32_t* class_tiles, const uint2* ranges, uint2* pairs, uint2* pairs_tmp,
                            uint32_t* big_hist, int max_tile_count, int index_bits, uint32_t* point_list, hipStream_t stream)
{
    int cc[FRG_SORT_CLASSES];
    for (int k = 0; k < FRG_SORT_CLASSES; k++) {
        if (class_count) cc[k] = (int)class_count[k];
        else {
            // estimate only: a quarter more than last time, never none (the lists are re-read on the device)
            const uint32_t h = grid_hint ? grid_hint[k] + grid_hint[k] / 4 + 8 : (uint32_t)T;
            cc[k] = (int)(h < (uint32_t)T ? h : (uint32_t)T);
            if (k == 4 && !pairs_tmp) cc[k] = 0;
        }
    }
    const int c0 = cc[0], c1 = cc[1], c2 = cc[2], c3 = cc[3], c4 = cc[4];
    const uint32_t* len = class_count_dev;
    if (T <= 0 || c0 + c1 + c2 + c3 + c4 == 0) return hipSuccess;
    thread_local SortStreams ss;
    const bool big = (c2 + c3 + c4) > 0;
    const bool forked = big && ss.ensure();
    hipStream_t s1 = stream, s2 = stream;
    hipError_t e;
    if (forked) {
        if ((e = hipEventRecord(ss.fork, stream)) != hipSuccess) return e;
        s1 = ss.side[0]; s2 = ss.side[1];
        if (c2 && (e = hipStreamWaitEvent(s1, ss.fork, 0)) != hipSuccess) return e;
        if ((c3 + c4) && (e = hipStreamWaitEvent(s2, ss.fork, 0)) != hipSuccess) return e;
    }
    // size classes: (0,512] 1 wave, (512,2048] 4 waves, (2048,4096] 8 waves, (4096,8192] 8 waves x 16 elements,
    // >8192 global ping-pong; longest-running classes first
    auto launch_global_class = [&]() -> hipError_t {
        if (!c4) return hipSuccess;
        if (!pairs_tmp || !big_hist) return hipErrorInvalidValue;
        const uint32_t* list = class_tiles + (size_t)4 * T;
        // strips of the longest list per workgroup row; when its length is only an estimate the kernels stride
        const int strips = ((max_tile_count > 0 ? max_tile_count : 4 * FRG_SORT_LDS_CAP) + BIG_STRIP - 1) / BIG_STRIP;
        const dim3 grid((strips + BIG_THREADS / 64 - 1) / (BIG_THREADS / 64), c4 < 4096 ? c4 : 4096);
        uint2 *src = pairs, *dst = pairs_tmp;
        const int index_passes = (index_bits + 7) / 8;
        for (int pass = 0; pass < index_passes + 4; pass++) {
            const bool by_index = pass < index_passes, last = pass == index_passes + 3;
            const int shift = 8 * (by_index ? pass : pass - index_passes);
            if (by_index) hipLaunchKernelGGL((big_count_kernel<true>), grid, dim3(BIG_THREADS), 0, s2, list, len + 4, ranges, src, shift, big_hist);
            else          hipLaunchKernelGGL((big_count_kernel<false>), grid, dim3(BIG_THREADS), 0, s2, list, len + 4, ranges, src, shift, big_hist);
            hipLaunchKernelGGL(big_offsets_kernel, dim3(grid.y), dim3(256), 0, s2, list, len + 4, ranges, big_hist);
            if (by_index)  hipLaunchKernelGGL((big_move_kernel<true, false>), grid, dim3(BIG_THREADS), 0, s2, list, len + 4, ranges, src, dst, point_list, shift, big_hist);
            else if (last) hipLaunchKernelGGL((big_move_kernel<false, true>), grid, dim3(BIG_THREADS), 0, s2, list, len + 4, ranges, src, dst, point_list, shift, big_hist);
            else           hipLaunchKernelGGL((big_move_kernel<false, false>), grid, dim3(BIG_THREADS), 0, s2, list, len + 4, ranges, src, dst, point_list, shift, big_hist);
            uint2* t = src; src = dst; dst = t;
        }
        return hipGetLastError();
    };
    // a known non-empty >8192 class runs longest and goes first; when its length is only an estimate
    // (usually zero tiles) it goes last on its stream: its 1024-thread workgroups would otherwise wait
    // for a free CU while the class queued behind them sits idle
    if (class_count && (e = launch_global_class()) != hipSuccess) return e;
    // (4096,8192]: 8 waves x 16 staged elements rather than 16 x 8 -- two workgroups fit a CU and
    // one's barrier stalls overlap the other's ranking (0.286 -> 0.256 ms at C3)
    if ((e = launch_lds_class<8, FRG_SORT_LDS_CAP>(c3, class_tiles + (size_t)3 * T, len + 3, ranges, pairs, point_list, s2)) != hipSuccess) return e;
    if (!class_count && (e = launch_global_class()) != hipSuccess) return e;
    if ((e = launch_lds_class<8, 4096>(c2, class_tiles + (size_t)2 * T, len + 2, ranges, pairs, point_list, s1)) != hipSuccess) return e;
    if ((e = launch_lds_class<4, 2048>(c1, class_tiles + (size_t)1 * T, len + 1, ranges, pairs, point_list, stream)) != hipSuccess) return e;
    if ((e = launch_lds_class<1, 512>(c0, class_tiles, len, ranges, pairs, point_list, stream)) != hipSuccess) return e;
    if (forked) {
        if (c2) {
            if ((e = hipEventRecord(ss.join[0], s1)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(stream, ss.join[0], 0)) != hipSuccess) return e;
        }
        if (c3 + c4) {
            if ((e = hipEventRecord(ss.join[1], s2)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(stream, ss.join[1], 0)) != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace frg
