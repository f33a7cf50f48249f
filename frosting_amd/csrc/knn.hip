// Mean squared distance to the three nearest neighbours of every point.
//
// Replaces simple-knn's distCUDA2 (gaussian_splatting/submodules/simple-knn/simple_knn.cu:64-222,
// spatial.cu:15-26), which initialises the Gaussians' scales (gaussian_model.py:134,
// frosting_model.py:530).  The value is fully specified -- for point p the three smallest
// |p - q|^2 over q != p, summed smallest first and divided by 3 -- so any exact search gives the
// reference's numbers; this one is organised for a wave64 machine:
//   1. bounding box by order-preserving integer atomics, 30-bit Morton codes, one rocPRIM radix sort
//      of (code, index) pairs (the only global sort in this library), points gathered into Morton order;
//   2. leaf boxes of KNN_LEAF consecutive sorted points with their bounds;
//   3. one query per lane, 256 queries per workgroup over the same stretch of the curve: the leaf table
//      is streamed through LDS in slabs, a leaf is opened when ANY lane's current third-best distance
//      reaches it (ballot), its points are staged in LDS once for the workgroup and scanned by the lanes
//      that need it.  The own and the neighbouring leaves go first, so the bound is tight before the sweep.
// Squared distances are evaluated as (dx*dx + dy*dy) + dz*dz without contraction (the order of
// simple_knn.cu:150-151), the result as (b0 + b1 + b2) / 3.0f (:197).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.h"

#pragma clang fp contract(off)

namespace frg {

#define KNN_LEAF 256
#define KNN_SLAB 1024          // leaf descriptors staged per sweep step (24 KB)

__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// bbox[0..2] = min (ordered ints), bbox[3..5] = max; initialised to 0xFFFFFFFF / 0 by the host memsets
__global__ void __launch_bounds__(256)
knn_bbox_kernel(int P, const float* __restrict__ pts, uint32_t* __restrict__ bbox)
{
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
        for (int c = 0; c < 3; c++) { const float v = pts[3 * i + c]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], d, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], d, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&bbox[c], f2ord(lo[c])); atomicMax(&bbox[3 + c], f2ord(hi[c])); }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256)
knn_morton_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ bbox, uint32_t* __restrict__ codes,
                  uint32_t* __restrict__ idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t q[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float lo = ord2f(bbox[c]), hi = ord2f(bbox[3 + c]);
        const float ext = hi - lo;
        const float t = ext > 0.f ? (pts[3 * i + c] - lo) / ext : 0.f;
        q[c] = (uint32_t)fminf(fmaxf(t * 1023.0f, 0.f), 1023.f);
    }
    codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    idx[i] = (uint32_t)i;
}

// points in Morton order (float4: x, y, z, original index bits) and the bounds of every leaf
__global__ void __launch_bounds__(KNN_LEAF)
knn_leaf_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order, float4* __restrict__ sorted,
                float* __restrict__ leaf_lo, float* __restrict__ leaf_hi)
{
    const int i = blockIdx.x * KNN_LEAF + threadIdx.x;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (i < P) {
        const uint32_t o = order[i];
        const float x = pts[3 * o], y = pts[3 * o + 1], z = pts[3 * o + 2];
        sorted[i] = make_float4(x, y, z, __uint_as_float(o));
        lo[0] = hi[0] = x; lo[1] = hi[1] = y; lo[2] = hi[2] = z;
    }
    __shared__ float red[6][KNN_LEAF / 64];
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], d, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], d, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[c][threadIdx.x >> 6] = lo[c]; red[3 + c][threadIdx.x >> 6] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float a = red[threadIdx.x][0], b = red[3 + threadIdx.x][0];
        for (int w = 1; w < KNN_LEAF / 64; w++) { a = fminf(a, red[threadIdx.x][w]); b = fmaxf(b, red[3 + threadIdx.x][w]); }
        leaf_lo[3 * blockIdx.x + threadIdx.x] = a;
        leaf_hi[3 * blockIdx.x + threadIdx.x] = b;
    }
}

__device__ __forceinline__ void keep3(float d, float* best)
{
    // simple_knn.cu:147-160: insertion into the ascending triple
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > d) { const float t = best[j]; best[j] = d; d = t; }
}

__device__ __forceinline__ float box_dist2(const float* lo, const float* hi, float x, float y, float z)
{
    const float dx = fmaxf(fmaxf(lo[0] - x, x - hi[0]), 0.f);
    const float dy = fmaxf(fmaxf(lo[1] - y, y - hi[1]), 0.f);
    const float dz = fmaxf(fmaxf(lo[2] - z, z - hi[2]), 0.f);
    return (dx * dx + dy * dy) + dz * dz;
}

// workgroup = KNN_LEAF threads = the queries of one leaf
__global__ void __launch_bounds__(KNN_LEAF)
knn_search_kernel(int P, int nleaf, const float4* __restrict__ sorted, const float* __restrict__ leaf_lo,
                  const float* __restrict__ leaf_hi, float* __restrict__ out)
{
    __shared__ float4 s_pts[KNN_LEAF];
    __shared__ float s_lo[KNN_SLAB * 3], s_hi[KNN_SLAB * 3];
    __shared__ uint32_t s_open[KNN_SLAB / 32];      // bitmap: leaves of the slab some query must still open
    const int me = blockIdx.x * KNN_LEAF + threadIdx.x;
    const bool live = me < P;
    const float4 q = live ? sorted[me] : make_float4(0.f, 0.f, 0.f, 0.f);
    float best[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f};   // FLT_MAX (:169)

    auto scan_leaf = [&](int leaf, bool need) {          // all threads call; `need` selects who scans
        __syncthreads();
        const int j = leaf * KNN_LEAF + threadIdx.x;
        s_pts[threadIdx.x] = j < P ? sorted[j] : make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);
        __syncthreads();
        if (need) {
            const int cnt = min(KNN_LEAF, P - leaf * KNN_LEAF);
            for (int k = 0; k < cnt; k++) {
                if (leaf * KNN_LEAF + k == me) continue;
                const float4 p = s_pts[k];
                const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                keep3((dx * dx + dy * dy) + dz * dz, best);
            }
        }
    };
    // own leaf and its two neighbours on the curve first: a tight bound before the sweep
    const int own = blockIdx.x;
    scan_leaf(own, live);
    if (own > 0) scan_leaf(own - 1, live);
    if (own + 1 < nleaf) scan_leaf(own + 1, live);

    for (int base = 0; base < nleaf; base += KNN_SLAB) {
        const int cnt = min(KNN_SLAB, nleaf - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 3; i += KNN_LEAF) { s_lo[i] = leaf_lo[base * 3 + i]; s_hi[i] = leaf_hi[base * 3 + i]; }
        if (threadIdx.x < KNN_SLAB / 32) s_open[threadIdx.x] = 0u;
        __syncthreads();
        // which leaves of this slab does ANY query of the workgroup still have to open?
        for (int l = 0; l < cnt; l++) {
            const int leaf = base + l;
            if (leaf >= own - 1 && leaf <= own + 1) continue;
            const bool need = live && !(box_dist2(s_lo + 3 * l, s_hi + 3 * l, q.x, q.y, q.z) > best[2]);
            if (__ballot(need) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&s_open[l >> 5], 1u << (l & 31));
        }
        __syncthreads();
        // open them in ascending order (workgroup-uniform walk over the bitmap); the bound is re-tested
        // per query with its current third-best distance
        for (int w = 0; w < (cnt + 31) / 32; w++) {
            uint32_t bits = s_open[w];
            while (bits) {
                const int l = w * 32 + __builtin_ctz(bits);
                bits &= bits - 1u;
                const bool need = live && !(box_dist2(s_lo + 3 * l, s_hi + 3 * l, q.x, q.y, q.z) > best[2]);
                scan_leaf(base + l, need);
            }
        }
    }
    if (live) out[__float_as_uint(q.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

size_t knn_workspace_bytes(int P)
{
    const size_t Pp = (size_t)(P > 0 ? P : 1), nleaf = (Pp + KNN_LEAF - 1) / KNN_LEAF;
    size_t sort_tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (uint32_t*)nullptr, Pp, 0, 30, (hipStream_t)0);
    size_t o = 256;                                  // bounding box
    o += 4 * align_up(Pp * 4, 256);                  // codes, codes sorted, indices, indices sorted
    o += align_up(Pp * 16, 256);                     // points in Morton order
    o += 2 * align_up(nleaf * 12, 256);              // leaf bounds
    o += align_up(sort_tmp, 256);
    return o;
}

hipError_t launch_knn(int P, const float* pts, float* out, char* ws, hipStream_t s)
{
    const size_t Pp = (size_t)P;
    const int nleaf = (P + KNN_LEAF - 1) / KNN_LEAF;
    size_t o = 0;
    uint32_t* bbox = (uint32_t*)(ws + o); o += 256;
    uint32_t* codes = (uint32_t*)(ws + o); o += align_up(Pp * 4, 256);
    uint32_t* codes_s = (uint32_t*)(ws + o); o += align_up(Pp * 4, 256);
    uint32_t* idx = (uint32_t*)(ws + o); o += align_up(Pp * 4, 256);
    uint32_t* idx_s = (uint32_t*)(ws + o); o += align_up(Pp * 4, 256);
    float4* sorted = (float4*)(ws + o); o += align_up(Pp * 16, 256);
    float* leaf_lo = (float*)(ws + o); o += align_up((size_t)nleaf * 12, 256);
    float* leaf_hi = (float*)(ws + o); o += align_up((size_t)nleaf * 12, 256);
    void* sort_tmp = ws + o;
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, codes, codes_s, idx, idx_s, Pp, 0, 30, s);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(bbox, 0xFF, 12, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(bbox + 3, 0x00, 12, s)) != hipSuccess) return e;
    const int rb = min(1024, (P + 255) / 256);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(rb), dim3(256), 0, s, P, pts, bbox);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, pts, bbox, codes, idx);
    if ((e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, codes, codes_s, idx, idx_s, Pp, 0, 30, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(knn_leaf_kernel, dim3(nleaf), dim3(KNN_LEAF), 0, s, P, pts, idx_s, sorted, leaf_lo, leaf_hi);
    hipLaunchKernelGGL(knn_search_kernel, dim3(nleaf), dim3(KNN_LEAF), 0, s, P, nleaf, sorted, leaf_lo, leaf_hi, out);
    return hipGetLastError();
}

}  // namespace frg
