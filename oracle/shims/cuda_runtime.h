// Build shim (test infrastructure, ours): lets the reference's CUDA sources
// under /root/reference compile with hipcc for gfx950 without being edited or
// copied into this repo. Maps the handful of CUDA runtime names they use.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#define cudaMemcpy hipMemcpy
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemset hipMemset
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaError_t hipError_t
#define cudaMalloc hipMalloc
#define cudaFree hipFree
