// A stand-in for the traffic a collective leaves beside a compute kernel (tools/combine_bench.py --side-copy): NWG workgroups of
// 256 threads copy `bytes` from src to dst with 16-byte accesses on the given stream -- few workgroups, as RCCL's channel kernels
// hold a handful of CUs -- so that the combine pass's slowdown beside incoming packets can be MEASURED on one GPU instead of
// assumed (frosting_amd.parallel.SLOTSUM_LOCAL_MS["overlap_slowdown"]).  Not part of the product library.
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC side_copy.hip -o libside_copy.so
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(256) side_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16, int rounds)
{
    for (int r = 0; r < rounds; r++)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int side_copy(const void* src, void* dst, size_t bytes, int workgroups, int rounds, void* stream)
{
    hipLaunchKernelGGL(side_copy_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, bytes / 16, rounds);
    return (int)hipGetLastError();
}
