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
#include "sort_lds.h"

#include <algorithm>
#include <atomic>

namespace frg {

// CAP = 8 * threads; tiles with lo < n <= CAP are handled by this instantiation (the host
// launches one instantiation per size class; workgroups whose tile is outside exit at once).
template <int NWAVES, int CAP>
__global__ void __launch_bounds__(NWAVES * 64, CAP <= 4096 ? 6 : 4)   // (80 registers: three 4096-entry workgroups per CU)
sort_tiles_lds_kernel(const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ list_len,
                      const uint2* __restrict__ ranges, const uint2* __restrict__ pairs, uint32_t* __restrict__ point_list)
{
    constexpr int SORT_ITEMS = CAP / (NWAVES * 64);   // elements staged per thread
    static_assert(CAP == NWAVES * 64 * SORT_ITEMS && SORT_ITEMS >= 1, "CAP must be a multiple of the workgroup size");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2* buf = reinterpret_cast<uint2*>(smem);
    uint32_t* whist = reinterpret_cast<uint32_t*>(buf + CAP);
    uint32_t* scratch = whist + NWAVES * 256;
    constexpr int NT = NWAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the grid normally covers exactly the tiles of this size class (one pass of this loop); when the
    // host only had an estimate of the list length, workgroups stride over the device-side list
    const uint32_t ntiles = *list_len;
    for (uint32_t item = blockIdx.x; item < ntiles; item += gridDim.x) {
        const int tile = (int)tile_list[item];
        const uint2 rg = ranges[tile];
        const int n = (int)(rg.y - rg.x);
        if (n > CAP) continue;
        int begin, end;
        wave_strip<NWAVES>(n, wave, begin, end);
        uint2 e[SORT_ITEMS];
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; it++) {
            const int i = begin + it * 64 + lane;
            e[it] = i < end ? load_pair_stream(pairs + rg.x + i) : make_uint2(0u, 0u);   // (read once: non-temporal)
        }
        sort_block_lds<NWAVES, SORT_ITEMS>(e, n, begin, end, buf, whist, scratch);   // (opens with a barrier: the previous tile's LDS contents are dead)
        for (int i = tid; i < n; i += NT) point_list[rg.x + i] = buf[i].y;
    }
}

// ---- lists of 8193 .. FRG_SORT_MID_MAX entries: sorted chunks + exact splitters -----------------------------
// Clustered scenes have tile lists of 10^4 .. 10^5 entries.  (depth, index) is a TOTAL order, so such
// a list may be cut into key ranges that are sorted independently:
//   big_plan          one workgroup: the chunks (8192 entries) of every long list -> work items; table space per list
//   big_chunk_sort    every chunk sorted in LDS by (depth, index) -> pairs_tmp                     (8 B read + written)
//   big_splitters     one workgroup per list: every 64th entry of each sorted chunk is a SAMPLE (n / 64 <= 8192 of
//                     them); the samples are sorted in LDS and every g-th one is a splitter.  Between two consecutive
//                     splitters lie at most (g + m) * 64 - m entries of the m chunks -- chunk c contributes fewer
//                     than 64 per sample it owns in the range, plus fewer than 64 -- so with g = 128 - m every bucket
//                     fits the LDS sort, WHATEVER the key distribution (no fallback, no recursion).  The exact bucket
//                     boundaries in every chunk (binary searches) go to the list's table, the buckets to a work queue.
//   big_bucket_sort   per bucket: its piece of every chunk gathered (m contiguous runs), sorted in LDS, written to
//                     point_list at the bucket's offset = number of entries below its lower splitter   (8 B read, 4 written)
// 28 bytes per entry and three passes of the LDS sort's instructions, against 7 global passes of the LSD form.
// (FRG_SORT_CHUNK, FRG_SORT_SAMPLE, FRG_SORT_MID_MAX and the BigPlan layout: frg_common.h)
#define BIG_NW 8
#define BIG_ITEMS (FRG_SORT_CHUNK / (BIG_NW * 64))

// (depth bits, index) <= (depth bits, index)
__device__ __forceinline__ bool pair_le(uint2 a, uint2 b) { return a.x < b.x || (a.x == b.x && a.y <= b.y); }

__global__ void __launch_bounds__(1024)
big_plan_kernel(const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ list_len, const uint2* __restrict__ ranges,
                uint32_t* __restrict__ plan_base, uint32_t R)
{
    __shared__ uint32_t wsum[2][16];
    __shared__ uint32_t carry[2];
    const BigPlan pl = BigPlan::carve(plan_base, R);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) carry[tid] = 0;
    if (tid == 0) { pl.hdr[1] = 0; pl.hdr[3] = 0; }
    __syncthreads();
    const uint32_t nl = *list_len;
    for (uint32_t base = 0; base < nl; base += 1024) {
        const uint32_t i = base + tid;
        uint32_t tile = 0, m = 0, nb = 0, n = 0;
        if (i < nl) {
            tile = tile_list[i];
            const uint2 rg = ranges[tile];
            n = rg.y - rg.x;
            if (n > (uint32_t)FRG_SORT_MID_MAX) { n = 0; pl.hdr[3] = 1; }     // the LSD kernels take it
            if (n > (uint32_t)FRG_SORT_CHUNK) {
                m = (n + FRG_SORT_CHUNK - 1) / FRG_SORT_CHUNK;
                const uint32_t ns = (m - 1) * (FRG_SORT_CHUNK / FRG_SORT_SAMPLE) + (n - (m - 1) * FRG_SORT_CHUNK) / FRG_SORT_SAMPLE;
                nb = ns / (FRG_SORT_CHUNK / FRG_SORT_SAMPLE - m) + 1;
            }
        }
        // two exclusive prefix sums over the lists: chunk items, table words
        uint32_t v[2] = {m, nb * m}, ex[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t inc = wave_incl_scan_dpp(v[k]);
            if (lane == 63) wsum[k][wave] = inc;
            ex[k] = inc - v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; k++) {
            uint32_t off = carry[k];
            for (int w = 0; w < wave; w++) off += wsum[k][w];
            ex[k] += off;
        }
        if (m) {
            pl.lists[i] = make_uint4(tile, ex[1], m, nb);
            for (uint32_t c = 0; c < m; c++) pl.chunks[ex[0] + c] = make_uint2(i, c);
        } else if (i < nl) {
            pl.lists[i] = make_uint4(tile, 0u, 0u, 0u);
        }
        __syncthreads();
        if (tid == 1023) { carry[0] = ex[0] + v[0]; carry[1] = ex[1] + v[1]; }
        __syncthreads();
    }
    if (tid == 0) { pl.hdr[0] = carry[0]; pl.hdr[2] = nl; }
}

__global__ void __launch_bounds__(BIG_NW * 64)
big_chunk_sort_kernel(const uint2* __restrict__ ranges, const uint2* __restrict__ pairs, uint2* __restrict__ sorted,
                      const uint32_t* __restrict__ plan_base, uint32_t R)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2* buf = reinterpret_cast<uint2*>(smem);
    uint32_t* whist = reinterpret_cast<uint32_t*>(buf + FRG_SORT_CHUNK);
    uint32_t* scratch = whist + BIG_NW * 256;
    const BigPlan pl = BigPlan::carve(const_cast<uint32_t*>(plan_base), R);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nitems = pl.hdr[0];
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const uint2 it2 = pl.chunks[item];
        const uint2 rg = ranges[pl.lists[it2.x].x];
        const uint32_t first = rg.x + it2.y * FRG_SORT_CHUNK;
        const int n = (int)min((uint32_t)FRG_SORT_CHUNK, rg.y - first);
        int begin, end;
        wave_strip<BIG_NW>(n, wave, begin, end);
        uint2 e[BIG_ITEMS];
#pragma unroll
        for (int it = 0; it < BIG_ITEMS; it++) {
            const int i = begin + it * 64 + lane;
            e[it] = i < end ? pairs[first + i] : make_uint2(0u, 0u);
        }
        sort_block_lds<BIG_NW, BIG_ITEMS>(e, n, begin, end, buf, whist, scratch);
        for (int i = tid; i < n; i += BIG_NW * 64) sorted[first + i] = buf[i];
    }
}

__global__ void __launch_bounds__(BIG_NW * 64)
big_splitters_kernel(const uint2* __restrict__ ranges, const uint2* __restrict__ sorted, uint32_t* __restrict__ plan_base, uint32_t R)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2* buf = reinterpret_cast<uint2*>(smem);
    uint32_t* whist = reinterpret_cast<uint32_t*>(buf + FRG_SORT_CHUNK);
    uint32_t* scratch = whist + BIG_NW * 256;
    __shared__ uint32_t bucket_sum[FRG_SORT_CHUNK / FRG_SORT_SAMPLE + 1];   // entries at or below splitter k, all chunks
    __shared__ uint32_t queue_base;
    const BigPlan pl = BigPlan::carve(plan_base, R);
    constexpr int NT = BIG_NW * 64, SPC = FRG_SORT_CHUNK / FRG_SORT_SAMPLE;   // samples per full chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nl = pl.hdr[2];
    for (uint32_t li = blockIdx.x; li < nl; li += gridDim.x) {
        const uint4 info = pl.lists[li];
        const int m = (int)info.z, nb = (int)info.w;
        if (m == 0) continue;                                   // workgroup-uniform
        const uint2 rg = ranges[info.x];
        const int n = (int)(rg.y - rg.x);
        const int ns = (m - 1) * SPC + (n - (m - 1) * FRG_SORT_CHUNK) / FRG_SORT_SAMPLE;
        const int g = SPC - m;                                  // (g + m) * 64 <= 8192: every bucket fits the LDS sort
        const uint2* src = sorted + rg.x;
        int begin, end;
        wave_strip<BIG_NW>(ns, wave, begin, end);
        uint2 e[BIG_ITEMS];
#pragma unroll
        for (int it = 0; it < BIG_ITEMS; it++) {
            const int t = begin + it * 64 + lane;
            // sample t: the last entry of the (t % SPC)-th group of 64 of chunk t / SPC
            e[it] = t < end ? src[(size_t)(t / SPC) * FRG_SORT_CHUNK + (size_t)(t % SPC + 1) * FRG_SORT_SAMPLE - 1] : make_uint2(0u, 0u);
        }
        sort_block_lds<BIG_NW, BIG_ITEMS>(e, ns, begin, end, buf, whist, scratch);
        for (int k = tid; k < nb; k += NT) bucket_sum[k] = 0;
        __syncthreads();
        // pos[k][c] = entries of chunk c at or below splitter k = buf[(k + 1) * g - 1]
        uint32_t* table = pl.tables + info.y;
        for (int q = tid; q < (nb - 1) * m; q += NT) {
            const int k = q / m, c = q - k * m;
            const uint2 key = buf[(k + 1) * g - 1];
            const uint2* ch = src + (size_t)c * FRG_SORT_CHUNK;
            int lo = 0, hi = min(FRG_SORT_CHUNK, n - c * FRG_SORT_CHUNK);     // first entry above the key
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (pair_le(ch[mid], key)) lo = mid + 1; else hi = mid;
            }
            table[q] = (uint32_t)lo;
            atomicAdd(&bucket_sum[k], (uint32_t)lo);
        }
        __syncthreads();
        if (tid == 0) { bucket_sum[nb - 1] = (uint32_t)n; queue_base = atomicAdd(&pl.hdr[1], (uint32_t)nb); }
        __syncthreads();
        for (int k = tid; k < nb; k += NT) {
            const uint32_t below = k ? bucket_sum[k - 1] : 0u;
            pl.buckets[queue_base + k] = make_uint4(li, (uint32_t)k, below, bucket_sum[k] - below);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(BIG_NW * 64)
big_bucket_sort_kernel(const uint2* __restrict__ ranges, const uint2* __restrict__ sorted, uint32_t* __restrict__ point_list,
                       uint32_t* __restrict__ plan_base, uint32_t R)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2* buf = reinterpret_cast<uint2*>(smem);
    uint32_t* whist = reinterpret_cast<uint32_t*>(buf + FRG_SORT_CHUNK);
    uint32_t* scratch = whist + BIG_NW * 256;
    __shared__ uint32_t piece_lo[FRG_SORT_MID_MAX / FRG_SORT_CHUNK], piece_start[FRG_SORT_MID_MAX / FRG_SORT_CHUNK + 1];
    const BigPlan pl = BigPlan::carve(plan_base, R);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nbuckets = pl.hdr[1];
    for (uint32_t item = blockIdx.x; item < nbuckets; item += gridDim.x) {
        const uint4 bk = pl.buckets[item];
        const uint4 info = pl.lists[bk.x];
        const int m = (int)info.z, nb = (int)info.w, k = (int)bk.y;
        const uint2 rg = ranges[info.x];
        const int ntot = (int)(rg.y - rg.x);
        int n = (int)bk.w;
        if (n > FRG_SORT_CHUNK) { if (tid == 0) pl.hdr[3] = 2; n = 0; }   // cannot happen (bound above); never write out of bounds
        const uint32_t* table = pl.tables + info.y;
        __syncthreads();                                       // the previous bucket's piece table is dead
        if (tid < 64) {                                        // m <= 64: one wave scans the piece lengths
            uint32_t lo = 0, hi = 0;
            if (tid < m) {
                lo = k ? table[(k - 1) * m + tid] : 0u;
                hi = k < nb - 1 ? table[k * m + tid] : (uint32_t)min(FRG_SORT_CHUNK, ntot - tid * FRG_SORT_CHUNK);
                piece_lo[tid] = lo;
            }
            const uint32_t inc = wave_incl_scan_dpp(hi - lo);
            if (tid < m) piece_start[tid] = inc - (hi - lo);
            if (tid == 63) piece_start[m] = inc;               // (lanes >= m add nothing)
        }
        __syncthreads();
        int begin, end;
        wave_strip<BIG_NW>(n, wave, begin, end);
        uint2 e[BIG_ITEMS];
#pragma unroll
        for (int it = 0; it < BIG_ITEMS; it++) {
            const int i = begin + it * 64 + lane;
            e[it] = make_uint2(0u, 0u);
            if (i < end) {
                int c = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int mid = c + step;
                    if (mid < m && piece_start[mid] <= (uint32_t)i) c = mid;
                }
                e[it] = sorted[rg.x + (size_t)c * FRG_SORT_CHUNK + piece_lo[c] + ((uint32_t)i - piece_start[c])];
            }
        }
        sort_block_lds<BIG_NW, BIG_ITEMS>(e, n, begin, end, buf, whist, scratch);
        for (int i = tid; i < n; i += BIG_NW * 64) point_list[rg.x + bk.z + i] = buf[i].y;
    }
}

// ---- lists longer than FRG_SORT_MID_MAX: multi-workgroup LSD radix sort in global memory ----------------
// (Round 2 sorted every list above the LDS capacity this way: 21 launches, one wave per 1024-element strip ranking
// its strip twice per digit -- 1.6 ms for the 8 M elements in the 139 long lists of the clustered scene.  It remains for
// lists of more than half a million entries, beyond what one workgroup's sample sort of the splitter path holds.)
// Every 1024-element strip of such a list is one WAVE's work: per 8-bit digit
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
        if (n <= FRG_SORT_MID_MAX) continue;
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
        if (n <= FRG_SORT_MID_MAX) continue;       // workgroup-uniform
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
        if (n <= FRG_SORT_MID_MAX) continue;
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
    const size_t lds = (size_t)CAP * 8 + NW * 1024 + FRG_SORT_SCRATCH_WORDS * 4;
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
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr}, plan_fork = nullptr, plan_done = nullptr;
    bool plan_on_side = false;   // the last big_plan_kernel went to side[1] (plan_done says when it is through)
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
        if (plan_fork) (void)hipEventDestroy(plan_fork);
        if (plan_done) (void)hipEventDestroy(plan_done);
        fork = nullptr; plan_fork = nullptr; plan_done = nullptr;
        device = -1;
        for (int i = 0; i < 2; i++) {
            if (hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&join[i], hipEventDisableTiming) != hipSuccess) return false;
        }
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&plan_fork, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&plan_done, hipEventDisableTiming) != hipSuccess) return false;
        device = dev;
        return true;
    }
};

// the three LDS-sorting kernels of the splitter path use the largest class's workgroup shape; their dynamic LDS
// attribute is set once per device
template <int WHICH, typename K>
static hipError_t allow_sort_lds(K kernel, size_t lds)
{
    static std::atomic<unsigned long long> attr_set{0};   // one per kernel (WHICH)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (attr_set.load() & bit) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) attr_set.fetch_or(bit);
    return e;
}

static thread_local SortStreams g_sort_streams;
int g_sort_heavy_on_caller = 1;   // frg_set_option("sort_heavy_on_caller")

// The long lists' plan only needs the scan's outputs (ranges, the class lists): launched on the sort's second side
// stream BEFORE the scatter is enqueued, its single workgroup's latency chain hides under the scatter.
// fork_mode: 0 fork here and launch | 1 fork only (the caller does not know yet whether there are long lists: it enqueues
// the scatter next and comes back with mode 2) | 2 launch behind the fork of an earlier mode-1 call.
hipError_t launch_sort_plan(int T, const uint32_t* class_count, const uint32_t* class_count_dev, const uint32_t* class_tiles,
                            const uint2* ranges, uint32_t* big_plan, uint32_t R, hipStream_t stream, int fork_mode)
{
    if (!big_plan || T <= 0 || (class_count && class_count[4] == 0)) return hipSuccess;
    SortStreams& ss = g_sort_streams;
    hipStream_t s2 = stream;
    hipError_t e;
    if (ss.ensure()) {
        if (fork_mode != 2 && (e = hipEventRecord(ss.plan_fork, stream)) != hipSuccess) return e;
        if (fork_mode == 1) return hipSuccess;
        if ((e = hipStreamWaitEvent(ss.side[1], ss.plan_fork, 0)) != hipSuccess) return e;
        s2 = ss.side[1];
    } else if (fork_mode == 1) return hipSuccess;
    hipLaunchKernelGGL(big_plan_kernel, dim3(1), dim3(1024), 0, s2, class_tiles + (size_t)4 * T, class_count_dev + 4, ranges, big_plan, R);
    ss.plan_on_side = s2 != stream;
    if (ss.plan_on_side && (e = hipEventRecord(ss.plan_done, s2)) != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t launch_tile_sort(int T, const uint32_t* class_count, const uint32_t* grid_hint, const uint32_t* class_count_dev,
                            const uint32_t* class_tiles, const uint2* ranges, uint2* pairs, uint2* pairs_tmp,
                            uint32_t* big_hist, uint32_t* big_plan, uint32_t R, int max_tile_count, int index_bits,
                            uint32_t* point_list, hipStream_t stream, bool skip_small)
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
    const int c0 = skip_small ? 0 : cc[0], c1 = cc[1], c2 = cc[2], c3 = cc[3], c4 = cc[4];
    const uint32_t* len = class_count_dev;
    if (T <= 0 || c0 + c1 + c2 + c3 + c4 == 0) return hipSuccess;
    SortStreams& ss = g_sort_streams;
    const bool big = (c2 + c3 + c4) > 0;
    const bool forked = big && ss.ensure();
    hipError_t e;
    // Four groups of work -- the long lists' chain of kernels, the (4096, 8192] class, the (2048, 4096] class, the two
    // small classes -- on three lanes: the caller's stream and two side streams.  A kernel on a side stream starts
    // 15-30 us after the fork event (cross-queue signalling) and the caller's stream resumes ~10 us after the last join.
    // Default: small classes on the caller's stream, (2048, 4096] on the first side stream, (4096, 8192] and the chain
    // on the second.  When long lists are KNOWN to exist their chain of four kernels runs longest by far: it then goes
    // on the caller's stream -- it starts right behind the scatter and the blend right behind it, the joins long
    // satisfied -- and the other groups are spread over the side streams by their work (clustered scene: sort stage
    // 0.478 -> 0.470 ms, step -0.02 ms).  Without long lists the same rule was measured SLOWER (C3: sort 0.204 ->
    // 0.209 ms, in-process A/B): the small classes then start late on a side stream instead of at once.
    hipStream_t lane[3] = {stream, stream, stream};
    int on[4] = {2, 2, 1, 0};                    // lane of: chain, (4096,8192], (2048,4096], small classes
    if (forked) { lane[1] = ss.side[0]; lane[2] = ss.side[1]; }
    if (forked && class_count && c4 > 0 && g_sort_heavy_on_caller) {
        const double w[4] = {c4 ? 1e30 : 0.0, c3 * 8192.0, c2 * 4096.0, c1 * 2048.0 + c0 * 512.0};
        int order[4] = {0, 1, 2, 3};
        for (int i = 1; i < 4; i++)
            for (int j = i; j > 0 && w[order[j]] > w[order[j - 1]]; j--) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
        double load[3] = {0.0, 0.0, 0.0};
        for (int i = 0; i < 4; i++) {
            const int g = order[i];
            int l = 0;
            if (i > 0) { l = 1; if (load[2] < load[1]) l = 2; if (load[0] < load[l]) l = 0; }
            on[g] = l;
            load[l] += w[g];
        }
    }
    const bool has[4] = {c4 > 0, c3 > 0, c2 > 0, c1 + c0 > 0};
    bool use_lane[3] = {false, false, false};
    for (int g = 0; g < 4; g++) if (has[g]) use_lane[on[g]] = true;
    const bool use_s1 = forked && use_lane[1], use_s2 = forked && use_lane[2];
    if (forked) {
        if ((e = hipEventRecord(ss.fork, stream)) != hipSuccess) return e;
        if (use_s1 && (e = hipStreamWaitEvent(lane[1], ss.fork, 0)) != hipSuccess) return e;
        if (use_s2 && (e = hipStreamWaitEvent(lane[2], ss.fork, 0)) != hipSuccess) return e;
    }
    const hipStream_t s4 = lane[on[0]], s3 = lane[on[1]], s2c = lane[on[2]], s10 = lane[on[3]];
    // the chain follows its plan (launch_sort_plan: on the second side stream, forked before the scatter)
    if (c4 && ss.plan_on_side && s4 != ss.side[1] && (e = hipStreamWaitEvent(s4, ss.plan_done, 0)) != hipSuccess) return e;
    const hipStream_t s2 = s4;     // (name used by the chain's launches below)
    // size classes: (0,512] 1 wave, (512,2048] 4 waves, (2048,4096] 8 waves, (4096,8192] 8 waves x 16 elements,
    // >8192 sorted chunks + splitters (beyond FRG_SORT_MID_MAX: global LSD passes); longest-running classes first
    auto launch_global_class = [&]() -> hipError_t {
        if (!c4) return hipSuccess;
        if (!pairs_tmp || !big_plan) return hipErrorInvalidValue;
        const uint32_t* list = class_tiles + (size_t)4 * T;
        {   // 8193 .. FRG_SORT_MID_MAX entries
            constexpr size_t lds = (size_t)FRG_SORT_CHUNK * 8 + BIG_NW * 1024 + FRG_SORT_SCRATCH_WORDS * 4;
            if ((e = allow_sort_lds<0>(big_chunk_sort_kernel, lds)) != hipSuccess) return e;
            if ((e = allow_sort_lds<1>(big_splitters_kernel, lds)) != hipSuccess) return e;
            if ((e = allow_sort_lds<2>(big_bucket_sort_kernel, lds)) != hipSuccess) return e;
            // work items are counted on the device; the grids are bounds (the kernels stride): chunks <= R / 8192 + lists,
            // buckets < R / 4064 + lists
            const unsigned chunks = (unsigned)std::min<size_t>((size_t)R / FRG_SORT_CHUNK + (size_t)c4, 2048);
            const unsigned buckets = (unsigned)std::min<size_t>((size_t)R / 4064 + (size_t)c4, 2048);
            // (big_plan_kernel: launch_sort_plan, before the scatter)
            hipLaunchKernelGGL(big_chunk_sort_kernel, dim3(chunks), dim3(BIG_NW * 64), lds, s2, ranges, pairs, pairs_tmp, big_plan, R);
            hipLaunchKernelGGL(big_splitters_kernel, dim3(c4 < 2048 ? c4 : 2048), dim3(BIG_NW * 64), lds, s2, ranges, pairs_tmp, big_plan, R);
            hipLaunchKernelGGL(big_bucket_sort_kernel, dim3(buckets), dim3(BIG_NW * 64), lds, s2, ranges, pairs_tmp, point_list, big_plan, R);
        }
        // beyond: only when the longest list is known to need it (deferred counters: not known not to)
        if (big_hist && (max_tile_count == 0 || max_tile_count > FRG_SORT_MID_MAX)) {
            // strips of the longest list per workgroup row; when its length is only an estimate the kernels stride
            const int strips = ((max_tile_count > 0 ? max_tile_count : 4 * FRG_SORT_MID_MAX) + BIG_STRIP - 1) / BIG_STRIP;
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
        }
        return hipGetLastError();
    };
    // within a lane: the longer-running group first.  (Sizes unknown: the chain's 1024-thread workgroups -- usually
    // of an empty list -- go behind the (4096, 8192] class, or they would wait for a free CU while that class sits
    // idle behind them.)
    if (class_count && (e = launch_global_class()) != hipSuccess) return e;
    // (4096,8192]: 8 waves x 16 staged elements rather than 16 x 8 -- two workgroups fit a CU and
    // one's barrier stalls overlap the other's ranking (0.286 -> 0.256 ms at C3)
    if ((e = launch_lds_class<8, FRG_SORT_LDS_CAP>(c3, class_tiles + (size_t)3 * T, len + 3, ranges, pairs, point_list, s3)) != hipSuccess) return e;
    if (!class_count && (e = launch_global_class()) != hipSuccess) return e;
    if ((e = launch_lds_class<8, 4096>(c2, class_tiles + (size_t)2 * T, len + 2, ranges, pairs, point_list, s2c)) != hipSuccess) return e;
    if ((e = launch_lds_class<4, 2048>(c1, class_tiles + (size_t)1 * T, len + 1, ranges, pairs, point_list, s10)) != hipSuccess) return e;
    if ((e = launch_lds_class<1, 512>(c0, class_tiles, len, ranges, pairs, point_list, s10)) != hipSuccess) return e;
    if (forked) {
        if (use_s1) {
            if ((e = hipEventRecord(ss.join[0], lane[1])) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(stream, ss.join[0], 0)) != hipSuccess) return e;
        }
        if (use_s2) {
            if ((e = hipEventRecord(ss.join[1], lane[2])) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(stream, ss.join[1], 0)) != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace frg
