// TEST INFRASTRUCTURE: C entry point around the reference's own SimpleKNN::knn
// (gaussian_splatting/submodules/simple-knn/simple_knn.cu:186-222, compiled where it lies by
// oracle/build_ref.sh).  points: P x float3 on the device, mean_dist2: P floats on the device.
#include <hip/hip_runtime.h>
#include "simple_knn.h"

extern "C" int ref_knn(int P, float* points, float* mean_dist2)
{
    SimpleKNN::knn(P, reinterpret_cast<float3*>(points), mean_dist2);
    return (int)hipDeviceSynchronize();
}
