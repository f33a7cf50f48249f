// Fused Adam step over the flat per-Gaussian parameter layout (the step
// right after the backward / the gradient exchange).
//
// Replaces torch.optim.Adam(groups, lr=0.0, eps=1e-15).step() as the reference configures it
// (frosting_scene/frosting_optimizer.py:74-121, gaussian_splatting/scene/gaussian_model.py:149-167:
// one parameter group per tensor, per-group learning rate, betas (0.9, 0.999), no weight decay, no
// amsgrad).  The eager optimizer runs ~6 elementwise kernels per group; here parameters, gradients and
// both moments live in one flat fp32 layout each (the gradient layout is the exchange buffer of
// frosting_amd/parallel.py) and ONE launch updates all groups: 16 bytes read + 12 written per
// element, the HBM floor of the operation.
//
// Arithmetic follows torch/optim/adam.py::_single_tensor_adam (non-capturable path):
//   m <- lerp(m, g, 1 - beta1);  v <- v * beta2 + (1 - beta2) * g * g
//   p <- p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with the bias corrections evaluated in double on the host, like the Python floats there.
#include "../../include/frosting_rasterizer.h"
#include "kernels.h"

namespace frg {

typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float* p)
{
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(float* p, float4 v) { __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4*>(p)); }

__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, float step_size, float w1,
                                          float beta2, float omb2, float inv_bc2_sqrt, float eps)
{
    // Tensor.lerp_(end, weight): weight < 0.5 ? self + weight * (end - self) : end - (end - self) * (1 - weight)
    const float d = g - m;
    m = w1 < 0.5f ? m + w1 * d : g - d * (1.0f - w1);
    v = v * beta2 + omb2 * g * g;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);
    return p;
}

// Everything the kernel needs to know about the segment an element lies in, selected with STATIC indices (an unrolled
// chain of selects over the at most eight segments): indexing the by-value argument structs with a computed k makes the
// compiler fetch them from the kernel-argument segment with vector loads, one dependent memory round trip per field.
struct SegInfo {
    long long begin;
    int k, period, head, width;
    float step, head_step;
    uint32_t magic;
};
__device__ __forceinline__ SegInfo seg_lookup(const AdamSegments& seg, const AdamRows& rows, long long i)
{
    SegInfo s{0, 0, seg.period[0], seg.head[0], rows.width[0], seg.step_size[0], seg.head_step_size[0], rows.magic[0]};
#pragma unroll
    for (int j = 1; j < FRG_ADAM_MAX_SEGMENTS; j++)
        if (j < seg.count && i >= seg.end[j - 1])
            s = SegInfo{seg.end[j - 1], j, seg.period[j], seg.head[j], rows.width[j], seg.step_size[j], seg.head_step_size[j], rows.magic[j]};
    return s;
}
__device__ __forceinline__ int seg_phase(const SegInfo& s, long long i)
{
    if (s.period <= 0) return 0;
    const unsigned long long off = (unsigned long long)(i - s.begin);
    return off < 0x100000000ull ? (int)((uint32_t)off % (uint32_t)s.period) : (int)(off % (unsigned long long)s.period);
}
__device__ __forceinline__ float seg_step(const SegInfo& s, int phase) { return (s.period > 0 && phase < s.head) ? s.head_step : s.step; }

// ROWS: a byte per Gaussian says whether it has a gradient (frg_backward_args::row_live); an unmarked Gaussian's gradient
// rows were never written -- they are ZERO by definition and are not read: at C3 six rows in seven, 0.6 of the 0.7 GB of
// gradients.  The moments decay and the parameter moves by its momentum exactly as with a stored zero.
// Is element i's gradient stored?  Its Gaussian = offset in the segment / elements per Gaussian -- a 32-bit division by a
// small constant as a multiplication by floor(2^32 / width) and one fix-up; the pad elements behind the last Gaussian of a
// segment have no gradient.
__device__ __forceinline__ bool row_is_live(const SegInfo& s, const AdamRows& rows, long long i)
{
    if (s.width <= 0) return true;
    const uint32_t off = (uint32_t)(i - s.begin), d = (uint32_t)s.width;
    uint32_t gi = off;
    if (d > 1u) {
        const uint32_t q = __umulhi(off, s.magic);
        gi = q + ((off - q * d) >= d ? 1u : 0u);
    }
    return gi < (uint32_t)rows.P && rows.live[gi] != 0;
}

// (the masked form; the dense step keeps its own kernel below: 46 registers, eight waves per SIMD)
template <bool ROWS>
__global__ void __launch_bounds__(256)
adam_step_rows_kernel(long long n, float* __restrict__ params, const float* __restrict__ grads, float* __restrict__ exp_avg,
                 float* __restrict__ exp_avg_sq, AdamSegments seg, float w1, float beta2, float omb2,
                 float inv_bc2_sqrt, float eps, float grad_scale, AdamRows rows)
{
    // 4 consecutive elements per thread (one 16-byte access per array); n4 = full groups of four
    const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long base = i4 * 4;
    if (base >= n) return;
    if (base + 4 <= n) {
        // a group of four may straddle a segment boundary or a period: one segment search and one modulo in the common
        // case, the phase then just counts up
        const SegInfo sa = seg_lookup(seg, rows, base), sb = seg_lookup(seg, rows, base + 3);
        const bool one_seg = sa.k == sb.k;
        // four streams in, three out, each touched once per step: non-temporal (nothing of them is worth a cache line)
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 p, m, v;
        if (ROWS) {
            // the mask bytes FIRST (their requests lead the queue: the gradient load that depends on them then waits for one
            // short read while the three streams behind it are in flight).  A group of four lies inside one Gaussian when
            // its segment's rows are a multiple of four elements long (the SH rows, 81 % of the elements: one look-up)
            bool l0, l1, l2, l3;
            if (one_seg && sa.width > 0 && (sa.width & 3) == 0) l0 = l1 = l2 = l3 = row_is_live(sa, rows, base);
            else if (one_seg) { l0 = row_is_live(sa, rows, base); l1 = row_is_live(sa, rows, base + 1); l2 = row_is_live(sa, rows, base + 2); l3 = row_is_live(sa, rows, base + 3); }
            else {
                l0 = row_is_live(sa, rows, base); l3 = row_is_live(sb, rows, base + 3);
                l1 = row_is_live(seg_lookup(seg, rows, base + 1), rows, base + 1); l2 = row_is_live(seg_lookup(seg, rows, base + 2), rows, base + 2);
            }
            p = ld_stream(params + base);
            m = ld_stream(exp_avg + base);
            v = ld_stream(exp_avg_sq + base);
            if (l0 | l1 | l2 | l3) {
                g = ld_stream(grads + base);
                g.x = l0 ? g.x : 0.0f; g.y = l1 ? g.y : 0.0f; g.z = l2 ? g.z : 0.0f; g.w = l3 ? g.w : 0.0f;
            }
        } else {
            p = ld_stream(params + base);
            g = ld_stream(grads + base);
            m = ld_stream(exp_avg + base);
            v = ld_stream(exp_avg_sq + base);
        }
        g.x *= grad_scale; g.y *= grad_scale; g.z *= grad_scale; g.w *= grad_scale;
        float s0, s1, s2, s3;
        if (one_seg) {
            int ph = seg_phase(sa, base);
            const int per = sa.period;
            s0 = seg_step(sa, ph); ph = (per > 0 && ph + 1 == per) ? 0 : ph + 1;
            s1 = seg_step(sa, ph); ph = (per > 0 && ph + 1 == per) ? 0 : ph + 1;
            s2 = seg_step(sa, ph); ph = (per > 0 && ph + 1 == per) ? 0 : ph + 1;
            s3 = seg_step(sa, ph);
        } else {
            const SegInfo s1i = seg_lookup(seg, rows, base + 1), s2i = seg_lookup(seg, rows, base + 2);
            s0 = seg_step(sa, seg_phase(sa, base)); s1 = seg_step(s1i, seg_phase(s1i, base + 1));
            s2 = seg_step(s2i, seg_phase(s2i, base + 2)); s3 = seg_step(sb, seg_phase(sb, base + 3));
        }
        adam_one(p.x, g.x, m.x, v.x, s0, w1, beta2, omb2, inv_bc2_sqrt, eps);
        adam_one(p.y, g.y, m.y, v.y, s1, w1, beta2, omb2, inv_bc2_sqrt, eps);
        adam_one(p.z, g.z, m.z, v.z, s2, w1, beta2, omb2, inv_bc2_sqrt, eps);
        adam_one(p.w, g.w, m.w, v.w, s3, w1, beta2, omb2, inv_bc2_sqrt, eps);
        st_stream(params + base, p);
        st_stream(exp_avg + base, m);
        st_stream(exp_avg_sq + base, v);
    } else {
        for (long long i = base; i < n; i++) {
            const SegInfo si = seg_lookup(seg, rows, i);
            float p = params[i], m = exp_avg[i], v = exp_avg_sq[i];
            const float gi = (!ROWS || row_is_live(si, rows, i)) ? grads[i] : 0.0f;
            adam_one(p, gi * grad_scale, m, v, seg_step(si, seg_phase(si, i)), w1, beta2, omb2, inv_bc2_sqrt, eps);
            params[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
        }
    }
}

__global__ void __launch_bounds__(256)
adam_step_kernel(long long n, float* __restrict__ params, const float* __restrict__ grads, float* __restrict__ exp_avg,
                 float* __restrict__ exp_avg_sq, AdamSegments seg, float w1, float beta2, float omb2,
                 float inv_bc2_sqrt, float eps, float grad_scale)
{
    // 4 consecutive elements per thread (one 16-byte access per array); n4 = full groups of four
    const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long base = i4 * 4;
    if (base >= n) return;
    // segment of element i, and -- where the segment has a periodic head -- i's phase in the period
    auto seg_of = [&](long long i) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < FRG_ADAM_MAX_SEGMENTS; j++)
            if (j < seg.count && i >= seg.end[j - 1]) k = j;
        return k;
    };
    auto phase_of = [&](long long i, int k) -> int {
        if (seg.period[k] <= 0) return 0;
        const unsigned long long off = (unsigned long long)(i - (k ? seg.end[k - 1] : 0));
        return off < 0x100000000ull ? (int)((uint32_t)off % (uint32_t)seg.period[k]) : (int)(off % (unsigned long long)seg.period[k]);
    };
    auto step_at = [&](int k, int phase) { return (seg.period[k] > 0 && phase < seg.head[k]) ? seg.head_step_size[k] : seg.step_size[k]; };
    auto step_of = [&](long long i) { const int k = seg_of(i); return step_at(k, phase_of(i, k)); };
    if (base + 4 <= n) {
        // four streams in, three out, each touched once per step: non-temporal (nothing of them is worth a cache line)
        float4 p = ld_stream(params + base);
        float4 g = ld_stream(grads + base);
        float4 m = ld_stream(exp_avg + base);
        float4 v = ld_stream(exp_avg_sq + base);
        g.x *= grad_scale; g.y *= grad_scale; g.z *= grad_scale; g.w *= grad_scale;
        // a group of four may straddle a segment boundary: per-element step size
        // a group of four may straddle a segment boundary or a period: one segment search and one
        // modulo in the common case, the phase then just counts up
        const int k0 = seg_of(base), k3 = seg_of(base + 3);
        float s0, s1, s2, s3;
        if (k0 == k3) {
            int ph = phase_of(base, k0);
            const int per = seg.period[k0];
            s0 = step_at(k0, ph); ph = (per > 0 && ph + 1 == per) ? 0 : ph + 1;
            s1 = step_at(k0, ph); ph = (per > 0 && ph + 1 == per) ? 0 : ph + 1;
            s2 = step_at(k0, ph); ph = (per > 0 && ph + 1 == per) ? 0 : ph + 1;
            s3 = step_at(k0, ph);
        } else {
            s0 = step_of(base); s1 = step_of(base + 1); s2 = step_of(base + 2); s3 = step_of(base + 3);
        }
        adam_one(p.x, g.x, m.x, v.x, s0, w1, beta2, omb2, inv_bc2_sqrt, eps);
        adam_one(p.y, g.y, m.y, v.y, s1, w1, beta2, omb2, inv_bc2_sqrt, eps);
        adam_one(p.z, g.z, m.z, v.z, s2, w1, beta2, omb2, inv_bc2_sqrt, eps);
        adam_one(p.w, g.w, m.w, v.w, s3, w1, beta2, omb2, inv_bc2_sqrt, eps);
        st_stream(params + base, p);
        st_stream(exp_avg + base, m);
        st_stream(exp_avg_sq + base, v);
    } else {
        for (long long i = base; i < n; i++) {
            float p = params[i], m = exp_avg[i], v = exp_avg_sq[i];
            adam_one(p, grads[i] * grad_scale, m, v, step_of(i), w1, beta2, omb2, inv_bc2_sqrt, eps);
            params[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
        }
    }
}

hipError_t launch_adam_step(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                            const AdamSegments& seg, float w1, float beta2, float omb2, float inv_bc2_sqrt, float eps,
                            float grad_scale, hipStream_t s, const AdamRows* rows)
{
    const long long groups = (n + 3) / 4;
    const long long blocks = (groups + 255) / 256;
    if (rows && rows->live)
        hipLaunchKernelGGL(adam_step_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, n, params, grads, exp_avg, exp_avg_sq, seg,
                           w1, beta2, omb2, inv_bc2_sqrt, eps, grad_scale, *rows);
    else
        hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n, params, grads, exp_avg, exp_avg_sq, seg,
                           w1, beta2, omb2, inv_bc2_sqrt, eps, grad_scale);
    return hipGetLastError();
}

}  // namespace frg
