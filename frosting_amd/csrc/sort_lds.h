// The LDS radix sort of one tile list by one workgroup (device functions shared by the tile-sort kernels of sort.hip and the
// forward blend's fused form for short lists, blend_impl.h): a stable wave64 ballot-ranked LSD radix sort on the depth bits,
// depth ties by ascending Gaussian index -- the reference's stable sort of index-ordered (tile, depth) keys
// (rasterizer_impl.cu:303-308).  Integer arithmetic only: the including translation unit's floating-point mode plays no part.
#pragma once
#include "frg_common.h"

namespace frg {

// 8-byte load with the non-temporal hint: the scatter's pairs are dead once their tile is sorted
__device__ __forceinline__ uint2 load_pair_stream(const uint2* p)
{
    typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
    const nt_u2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(p));
    return make_uint2(v.x, v.y);
}

// Lanes of the wave holding the same 8-bit digit as this lane (invalid lanes excluded).  Per bit: the lane's bit
// sign-extended to a mask sb (v_bfe_i32), one ballot, and peers &= ~(ballot ^ sb) per 32-bit half (v_xnor + v_and):
// six vector instructions.  (`peers &= bit ? m : ~m` on 64-bit values compiled to nine: the ranking is the sort's
// instruction-bound inner loop.)
// `width` bits per digit (wave-uniform): the digits of a pass only have as many bits as the key range needs -- the ranking
// costs four to six vector instructions per BIT.  Rolled, two bits per trip: their compare -> ballot -> mask chains are
// independent, so the second hides the first's latency; one unrolled copy per digit width cost 20-40 registers.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid, int width = 8)
{
    const uint64_t v = __builtin_amdgcn_ballot_w64(valid);
    uint32_t plo = (uint32_t)v, phi = (uint32_t)(v >> 32);
    int b = 0;
#pragma unroll 1
    for (; b + 1 < width; b += 2) {
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);       // 0 or 0xFFFFFFFF
        const uint32_t s1 = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)(b + 1), 1u);
        const uint64_t m0 = __builtin_amdgcn_ballot_w64(s0 != 0u);
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(s1 != 0u);
        plo &= ~((uint32_t)m0 ^ s0) & ~((uint32_t)m1 ^ s1);
        phi &= ~((uint32_t)(m0 >> 32) ^ s0) & ~((uint32_t)(m1 >> 32) ^ s1);
    }
    if (b < width) {
        const uint32_t sb = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);
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

// Depth ties: order equal-depth runs by ascending index (the stable-sort tie rule).  Runs of up
// to FRG_TIE_RUN entries are put in order by insertion, each by the thread that finds its first entry; a longer run
// (coplanar scenes: thousands of equal depths) raises the flag and the caller re-sorts on the index first and the depth
// again, which LSD stability turns into (depth, index).  The criterion is the LENGTH of the runs, not their number: the
// 8192-entry chunks of a tight cluster hold ~150 equal-depth pairs each (float depths 4 +- 0.03 take 4e5 values), and
// counting them (round 2: more than 32 -> re-sort) sent every chunk through nine radix passes instead of three.
#define FRG_TIE_RUN 8
template <typename PtrT>
__device__ __forceinline__ bool fix_short_ties(PtrT a, int n, int nthreads, uint32_t* scratch)
{
    if (threadIdx.x == 0) scratch[260] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n - 1; i += nthreads) {
        const uint32_t k = a[i].x;
        if (a[i + 1].x != k) continue;
        if (i > 0 && a[i - 1].x == k) continue;  // not the run start
        int j = i + 1;
        while (j + 1 < n && j - i < FRG_TIE_RUN && a[j + 1].x == k) j++;
        if (j - i >= FRG_TIE_RUN) { scratch[260] = 1; continue; }
        for (int p = i + 1; p <= j; p++) {  // insertion sort on .y within [i, j]
            const uint2 v = a[p];
            int q = p - 1;
            while (q >= i && a[q].y > v.y) { a[q + 1] = a[q]; q--; }
            a[q + 1] = v;
        }
    }
    __syncthreads();
    const bool long_run = scratch[260] != 0;
    __syncthreads();
    return long_run;
}

// ---- LDS classes: register-staged, in-place passes --------------------------------
// Each thread owns CAP / threads (8 or 16) elements of its wave's contiguous strip.
// A pass ranks them from REGISTERS (no LDS reads of the data), scatters them into the
// single LDS buffer and reloads its strip: one 8-byte buffer instead of a ping-pong pair,
// so twice the workgroups fit per CU and different size classes can share a CU.
template <int NWAVES, int SORT_ITEMS, bool BY_INDEX>
__device__ __forceinline__ bool radix_pass_regs(uint2 (&e)[SORT_ITEMS], int n, int begin, int end, int shift, int width, int pass,
                                                uint32_t kmin, uint2* buf, uint32_t* whist, uint32_t* scratch)
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
        const uint32_t d = (((BY_INDEX ? e[it].y : e[it].x) - kmin) >> shift) & ((1u << width) - 1u);
        const uint64_t peers = match_digit(d, valid, width);
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
        if (run == (uint32_t)n) scratch[256 + pass] = 1;
    }
    __syncthreads();   // (every thread also holds its elements in registers by now: the buffer may be overwritten)
    if (scratch[256 + pass] != 0) return false;
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

// Smallest key (depth bits, or Gaussian index) of the workgroup's elements and the number of bits of (largest -
// smallest): the passes sort key - smallest, whose high bits are zero, in ceil(bits / 8) digits of just enough
// bits each.  The depths of one tile span a fraction of the float range -- 25 bits at C3 (depths 1 .. 7), 19 in a tight
// cluster -- so the ranking, six vector instructions per key BIT and the sort's instruction-bound part, runs on 25 bits
// instead of 32, and a range of 16 bits or less takes two passes.  scratch[264 .. 264 + 2 * 16): per-wave min / max.
template <int NWAVES, int SORT_ITEMS, bool BY_INDEX>
__device__ __forceinline__ void key_range(const uint2 (&e)[SORT_ITEMS], int begin, int end, uint32_t* scratch, uint32_t& kmin, int& nbits)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; it++) {
        const uint32_t k = BY_INDEX ? e[it].y : e[it].x;
        if (begin + it * 64 + lane < end) { lo = min(lo, k); hi = max(hi, k); }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64)); }
    __syncthreads();                       // earlier readers of the scratch words are done
    if (lane == 0) { scratch[264 + wave] = lo; scratch[264 + 16 + wave] = hi; }
    __syncthreads();
    lo = 0xFFFFFFFFu; hi = 0u;
#pragma unroll
    for (int w = 0; w < NWAVES; w++) { lo = min(lo, scratch[264 + w]); hi = max(hi, scratch[264 + 16 + w]); }
    kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);          // (the same in every lane: scalar registers)
    const uint32_t range = (uint32_t)__builtin_amdgcn_readfirstlane((int)(hi - lo));
    nbits = hi >= lo ? 32 - __builtin_clz(range | 1u) : 0;              // no element at all: nothing to sort
    if (range == 0u) nbits = 0;
}

// the passes of one key: ceil(nbits / 8) digits, the first (nbits % passes) of them one bit wider
template <int NWAVES, int SORT_ITEMS, bool BY_INDEX>
__device__ __forceinline__ bool radix_passes(uint2 (&e)[SORT_ITEMS], int n, int begin, int end, uint32_t kmin, int nbits,
                                             uint2* buf, uint32_t* whist, uint32_t* scratch)
{
    const int npass = (nbits + 7) >> 3;
    bool moved = false;
    int shift = 0;
#pragma unroll 1
    for (int pass = 0; pass < npass; pass++) {
        const int width = nbits / npass + (pass < nbits % npass ? 1 : 0);
        moved |= radix_pass_regs<NWAVES, SORT_ITEMS, BY_INDEX>(e, n, begin, end, shift, width, pass, kmin, buf, whist, scratch);
        shift += width;
    }
    return moved;
}

// Sorts the workgroup's n <= NWAVES * 64 * SORT_ITEMS elements by (depth bits, index); e[] holds the calling thread's
// elements of its wave's strip (element begin + it * 64 + lane in e[it]).  On return buf[0, n) holds the sorted
// pairs (and every thread is past a barrier behind the last write).
template <int NWAVES, int SORT_ITEMS>
__device__ __forceinline__ void sort_block_lds(uint2 (&e)[SORT_ITEMS], int n, int begin, int end, uint2* buf, uint32_t* whist,
                                               uint32_t* scratch)
{
    constexpr int NT = NWAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t kmin;
    int nbits;
    key_range<NWAVES, SORT_ITEMS, false>(e, begin, end, scratch, kmin, nbits);
    if (tid < 4) scratch[256 + tid] = 0;
    __syncthreads();
    bool in_lds = false;
    if (n > 1) in_lds = radix_passes<NWAVES, SORT_ITEMS, false>(e, n, begin, end, kmin, nbits, buf, whist, scratch);
    if (!in_lds) {   // nothing moved (n == 1 or all keys equal): materialise the strip for the steps below
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; it++) {
            const int i = begin + it * 64 + lane;
            if (i < end) buf[i] = e[it];
        }
        __syncthreads();
    }
    if (n > 1) {
        if (fix_short_ties<uint2*>(buf, n, NT, scratch)) {
            // long runs of equal depths (coplanar scenes): order by index, then by depth again -- LSD
            // stability turns that into (depth, index).  (e[] still holds the strips as the last pass left them.)
            uint32_t imin;
            int ibits;
            key_range<NWAVES, SORT_ITEMS, true>(e, begin, end, scratch, imin, ibits);
            if (tid < 4) scratch[256 + tid] = 0;      // (the first pass has a barrier before the flags are used)
            radix_passes<NWAVES, SORT_ITEMS, true>(e, n, begin, end, imin, ibits, buf, whist, scratch);
            __syncthreads();                          // every thread is past the last pass's flag
            if (tid < 4) scratch[256 + tid] = 0;
            radix_passes<NWAVES, SORT_ITEMS, false>(e, n, begin, end, kmin, nbits, buf, whist, scratch);
        }
    }
}

// the strip of wave w in a block of n elements: contiguous (keeps every pass stable), a multiple of 64, at most 64 * SORT_ITEMS
template <int NWAVES>
__device__ __forceinline__ void wave_strip(int n, int wave, int& begin, int& end)
{
    const int strip = ((n + NWAVES - 1) / NWAVES + 63) & ~63;
    begin = wave * strip;
    end = min(n, begin + strip);
}

#define FRG_SORT_SCRATCH_WORDS (264 + 32)   // 256 digit totals, 4 uniform-pass flags, tie counter, per-wave OR / AND of the keys

}  // namespace frg
