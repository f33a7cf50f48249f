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

// Lanes of the wave holding the same 8-bit digit as this lane (invalid lanes excluded).
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid)
{
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask, int lane)
{
    return (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// One stable counting pass on digit `shift` from src to dst (n elements).
// whist: [NWAVES][256] per-wave digit counters, wave-private during the sweeps.
// Returns false (and moves nothing) when every key has the same digit.
template <int NWAVES, bool BY_INDEX, typename PtrT>
__device__ __forceinline__ bool radix_pass(PtrT src, PtrT dst, int n, int shift, uint32_t* whist, uint32_t* scratch)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = NWAVES * 64;
    for (int i = tid; i < NWAVES * 256; i += NT) whist[i] = 0;
    __syncthreads();
    // contiguous strip per wave (keeps the pass stable), multiple of 64
    const int strip = ((n + NWAVES - 1) / NWAVES + 63) & ~63;
    const int begin = wave * strip, end = min(n, begin + strip);
    uint32_t* myhist = whist + wave * 256;
    for (int i = begin; i < end; i += 64) {
        const bool valid = i + lane < end;
        const uint32_t key = valid ? (BY_INDEX ? src[i + lane].y : src[i + lane].x) : 0u;
        const uint32_t d = (key >> shift) & 255u;
        const uint64_t peers = match_digit(d, valid);
        if (valid && lanes_below(peers, lane) == 0) myhist[d] += (uint32_t)__popcll(peers);
    }
    __syncthreads();
    // digit totals, exclusive over waves then over digits
    if (tid < 256) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; w++) {
            const uint32_t c = whist[w * 256 + tid];
            whist[w * 256 + tid] = run;
            run += c;
        }
        scratch[tid] = run;  // total for digit tid
    }
    __syncthreads();
    if (tid < 256 && scratch[tid] == (uint32_t)n) scratch[256] = 1;  // one digit holds everything
    __syncthreads();
    const bool uniform = scratch[256] != 0;
    __syncthreads();
    if (uniform) {
        if (tid == 0) scratch[256] = 0;
        __syncthreads();
        return false;
    }
    if (tid < 64) {  // exclusive scan of 256 totals by one wave, 4 per lane
        uint32_t v0 = scratch[4 * lane], v1 = scratch[4 * lane + 1], v2 = scratch[4 * lane + 2], v3 = scratch[4 * lane + 3];
        uint32_t s = v0 + v1 + v2 + v3, inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        uint32_t ex = inc - s;
        scratch[4 * lane] = ex; scratch[4 * lane + 1] = ex + v0;
        scratch[4 * lane + 2] = ex + v0 + v1; scratch[4 * lane + 3] = ex + v0 + v1 + v2;
    }
    __syncthreads();
    if (tid < 256) {
        const uint32_t base = scratch[tid];
#pragma unroll
        for (int w = 0; w < NWAVES; w++) whist[w * 256 + tid] += base;
    }
    __syncthreads();
    for (int i = begin; i < end; i += 64) {
        const bool valid = i + lane < end;
        uint2 e = make_uint2(0u, 0u);
        if (valid) e = src[i + lane];
        const uint32_t d = ((BY_INDEX ? e.y : e.x) >> shift) & 255u;
        const uint64_t peers = match_digit(d, valid);
        const uint32_t rank = lanes_below(peers, lane);
        uint32_t pos = 0;
        if (valid) pos = myhist[d] + rank;
        // all lanes have read myhist before any leader bumps it (same wave, LDS ops in order,
        // but make the dependence explicit for the compiler)
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) myhist[d] += (uint32_t)__popcll(peers);
        if (valid) dst[pos] = e;
    }
    __syncthreads();
    return true;
}

// Depth ties: order equal-depth runs by ascending index (the stable-sort tie rule,
// SURVEY Appendix A-7).  count_ties() returns the number of adjacent equal-depth
// pairs; a handful are fixed by insertion (fix_ties), many (coplanar scenes) by
// re-sorting on the index first and the depth again, which LSD stability turns
// into (depth, index) order.
template <typename PtrT>
__device__ __forceinline__ int count_ties(PtrT a, int n, int nthreads, uint32_t* scratch)
{
    if (threadIdx.x == 0) scratch[257] = 0;
    __syncthreads();
    uint32_t c = 0;
    for (int i = threadIdx.x; i < n - 1; i += nthreads) c += (a[i].x == a[i + 1].x) ? 1u : 0u;
    if (c) atomicAdd(&scratch[257], c);
    __syncthreads();
    const int r = (int)scratch[257];
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

// Full (depth, index) ordering of one segment; returns the buffer holding the result.
template <int NWAVES, typename PtrT>
__device__ __forceinline__ PtrT sort_segment(PtrT src, PtrT dst, int n, uint32_t* whist, uint32_t* scratch)
{
    constexpr int NT = NWAVES * 64;
#pragma unroll 1
    for (int pass = 0; pass < 4; pass++) {
        if (radix_pass<NWAVES, false, PtrT>(src, dst, n, 8 * pass, whist, scratch)) { PtrT t = src; src = dst; dst = t; }
    }
    const int ties = count_ties<PtrT>(src, n, NT, scratch);
    if (ties == 0) return src;
    if (ties <= 32) { fix_ties<PtrT>(src, n, NT); return src; }
#pragma unroll 1
    for (int pass = 0; pass < 4; pass++) {
        if (radix_pass<NWAVES, true, PtrT>(src, dst, n, 8 * pass, whist, scratch)) { PtrT t = src; src = dst; dst = t; }
    }
#pragma unroll 1
    for (int pass = 0; pass < 4; pass++) {
        if (radix_pass<NWAVES, false, PtrT>(src, dst, n, 8 * pass, whist, scratch)) { PtrT t = src; src = dst; dst = t; }
    }
    return src;
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
        if (valid && rank == 0) myhist[d] += cnt;
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
        if (run == (uint32_t)n) scratch[256] = 1;   // one digit holds everything: nothing to move
    }
    __syncthreads();
    const bool uniform = scratch[256] != 0;
    __syncthreads();
    if (uniform) {
        if (tid == 0) scratch[256] = 0;
        __syncthreads();
        return false;
    }
    if (tid < 64) {  // exclusive scan of the 256 totals by one wave, 4 per lane
        uint32_t v0 = scratch[4 * lane], v1 = scratch[4 * lane + 1], v2 = scratch[4 * lane + 2], v3 = scratch[4 * lane + 3];
        uint32_t s = v0 + v1 + v2 + v3, inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        uint32_t ex = inc - s;
        scratch[4 * lane] = ex; scratch[4 * lane + 1] = ex + v0;
        scratch[4 * lane + 2] = ex + v0 + v1; scratch[4 * lane + 3] = ex + v0 + v1 + v2;
    }
    __syncthreads();
    for (int dg = tid; dg < 256; dg += NT) {
        const uint32_t base = scratch[dg];
#pragma unroll
        for (int w = 0; w < NWAVES; w++) whist[w * 256 + dg] += base;
    }
    __syncthreads();   // every thread holds its elements in registers: the buffer may be overwritten
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; it++) {
        if (begin + it * 64 >= end) break;        // wave-uniform
        const bool valid = begin + it * 64 + lane < end;
        const uint32_t rank = meta[it] & 255u, cnt = (meta[it] >> 8) & 255u, d = meta[it] >> 16;
        uint32_t pos = 0;
        if (valid) pos = myhist[d] + rank;
        __builtin_amdgcn_wave_barrier();          // all lanes read the cursor before a leader bumps it
        if (valid && rank == 0) myhist[d] += cnt;
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
    uint32_t* scratch = whist + NWAVES * 256;  // 260 words
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
        if (tid == 0) scratch[256] = 0;
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
#pragma unroll 1
                for (int pass = 0; pass < 4; pass++) radix_pass_regs<NWAVES, SORT_ITEMS, true>(e, n, begin, end, 8 * pass, buf, whist, scratch);
#pragma unroll 1
                for (int pass = 0; pass < 4; pass++) radix_pass_regs<NWAVES, SORT_ITEMS, false>(e, n, begin, end, 8 * pass, buf, whist, scratch);
            } else if (ties > 0) {
                fix_ties<uint2*>(buf, n, NT);
            }
        }
        for (int i = tid; i < n; i += NT) point_list[rg.x + i] = buf[i].y;
    }
}

// Fallback for tile lists longer than the LDS capacity: same passes, ping-pong
// in global memory (pairs <-> pairs_tmp); only histograms live in LDS.
template <int NWAVES>
__global__ void __launch_bounds__(NWAVES * 64)
sort_tiles_global_kernel(const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ list_len,
                         const uint2* __restrict__ ranges, uint2* pairs, uint2* pairs_tmp, uint32_t* __restrict__ point_list)
{
    __shared__ uint32_t whist[NWAVES * 256];
    __shared__ uint32_t scratch[260];
    constexpr int NT = NWAVES * 64;
    const int tid = threadIdx.x;
    const uint32_t ntiles = *list_len;
    for (uint32_t item = blockIdx.x; item < ntiles; item += gridDim.x) {
        __syncthreads();
        const int tile = (int)tile_list[item];
        const uint2 rg = ranges[tile];
        const int n = (int)(rg.y - rg.x);
        if (tid == 0) scratch[256] = 0;
        __syncthreads();
        uint2* src = sort_segment<NWAVES, uint2*>(pairs + rg.x, pairs_tmp + rg.x, n, whist, scratch);
        for (int i = tid; i < n; i += NT) point_list[rg.x + i] = src[i].y;
    }
}

// Host-side launcher (called from api.hip)
template <int NW, int CAP>
static hipError_t launch_lds_class(int count, const uint32_t* tile_list, const uint32_t* list_len, const uint2* ranges,
                                   const uint2* pairs, uint32_t* point_list, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    const size_t lds = (size_t)CAP * 8 + NW * 1024 + 260 * 4;
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
                            uint32_t* point_list, hipStream_t stream)
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
        if (!pairs_tmp) return hipErrorInvalidValue;
        hipLaunchKernelGGL((sort_tiles_global_kernel<16>), dim3(c4), dim3(1024), 0, s2, class_tiles + (size_t)4 * T, len + 4, ranges, pairs, pairs_tmp, point_list);
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
