// Microbenchmark: issue cost of the instruction kinds the blend kernels are made of, on gfx950 (wave64).
// 2048 workgroups x 256 threads, 8 independent chains per thread; reports cycles per wave-instruction per SIMD.
// hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHAINS 8
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b)
{
    float x[CHAINS]; f2 y[CHAINS];
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(a, b, a, b);
    __syncthreads();
    for (int i = 0; i < CHAINS; i++) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f2{x[i], x[i] + 0.5f}; }
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
            if (MODE == 2) x[i] = x[i] * a;
            if (MODE == 3) x[i] = x[i] + b;
            if (MODE == 4) x[i] = __builtin_amdgcn_exp2f(x[i]);
            if (MODE == 5) x[i] = __builtin_amdgcn_rcpf(x[i]);
            if (MODE == 6) x[i] = fminf(x[i], b);
            if (MODE == 7) x[i] = dpp_add<0xB1, 0xf>(x[i]);       // quad_perm
            if (MODE == 8) x[i] = dpp_add<0x140, 0xf>(x[i]);      // row_mirror
            if (MODE == 9) x[i] = dpp_add<0x142, 0xa>(x[i]);      // row_bcast15
            if (MODE == 10) { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[i]), __float_as_uint(x[(i + 1) % CHAINS]), false, false); x[i] = __uint_as_float(r[0]); }
            if (MODE == 11) x[i] = (x[i] > b) ? x[i] * a : x[i];   // v_cmp + v_cndmask (+ mul)
            if (MODE == 12) x[i] = __shfl_xor(x[i], 16, 64);      // ds_bpermute / swizzle
            if (MODE == 13) x[i] += lds[(it + i) & 255].x;        // uniform-address ds_read + add
            if (MODE == 14) { if (__ballot(x[i] > b) == 0ull) x[i] += a; }   // v_cmp + s_cbranch (uniform)
            if (MODE == 15) x[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x[i])) + 1u);
        }
    }
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* out)
{
    const int iters = 2048, blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 /*waves*/ * iters * CHAINS;   // wave-level operations
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms  %6.2f cycles per wave-op per SIMD (2.4 GHz assumed)\n", name, ms, cyc / (insts / 1024.0));
}
int main()
{
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<2>("v_mul_f32", out); run<3>("v_add_f32", out);
    run<4>("v_exp_f32", out); run<5>("v_rcp_f32", out); run<6>("v_min_f32", out);
    run<7>("v_add_f32_dpp quad_perm", out); run<8>("v_add_f32_dpp row_mirror", out); run<9>("v_add_f32_dpp row_bcast15", out);
    run<10>("v_permlane32_swap", out); run<11>("v_cmp+v_cndmask+v_mul", out); run<12>("__shfl_xor 16", out);
    run<13>("ds_read uniform + add", out); run<14>("v_cmp+ballot branch", out); run<15>("readfirstlane+s_add+mov", out);
    return 0;
}
