/* oracle/ref_cuda_names.h -- TEST INFRASTRUCTURE (oracle/build_ref.py): the CUDA runtime names that
 * /root/reference/PointCloud/openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu uses, mapped to HIP so that hipcc compiles
 * the reference file unmodified, in place. */
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString
using std::max;
using std::min;
