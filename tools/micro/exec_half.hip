// Microbenchmark: does a wave64 VALU instruction cost less when one 32-lane half of EXEC is empty?  (gfx950 executes a
// wave64 instruction as two passes of 32 lanes.)  Chains of v_fma_f32 / v_exp_f32 under four lane masks: all 64 lanes,
// the lower 32, the even lanes, one lane.
// hipcc --offload-arch=gfx950 -O3 exec_half.hip -o exec_half && ./exec_half
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS 8
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b, unsigned long long mask)
{
    float x[CHAINS];
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < CHAINS; i++) x[i] = threadIdx.x * 1e-3f + i;
    if ((mask >> lane) & 1ull) {           // one divergent region around the whole loop: EXEC = mask inside
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < CHAINS; i++) {
                if (OP == 0) x[i] = __builtin_fmaf(x[i], a, b);
                if (OP == 1) x[i] = __builtin_amdgcn_exp2f(x[i]);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, float* out, unsigned long long mask)
{
    const int iters = 2048, blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f, mask);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f, mask);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * CHAINS;
    printf("%-12s mask %016llx  %8.3f ms  %6.2f cycles per wave-op per SIMD (2.4 GHz assumed)\n", name, mask, ms, ms * 1e-3 * 2.4e9 / (insts / 1024.0));
}
int main()
{
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFFFFF00000000ull, 0x5555555555555555ull, 1ull};
    for (unsigned long long m : masks) run<0>("v_fma_f32", out, m);
    for (unsigned long long m : masks) run<1>("v_exp_f32", out, m);
    return 0;
}
