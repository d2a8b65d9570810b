// cu_hog -- test / measurement infrastructure, not product: a kernel that HOLDS `n_cus` compute units for `usec` microseconds, the
// way an RCCL all-reduce kernel holds the CUs of its channels while a gradient bucket is on the xGMI ring.  One workgroup per CU:
// each asks for the whole 160 KiB of LDS, so no second copy and none of the library's matrix kernels (128 KiB of LDS per
// workgroup) can share the CU; workgroup b lands on XCD b % 8, i.e. the held CUs spread over the XCDs as RCCL's channels do.
// The workgroup only sleeps on the 100 MHz constant clock -- no memory traffic, negligible power: what is rehearsed is the
// LOSS OF CUs to a communication kernel ("does the non-claiming wgrad kernel stall behind RCCL's CUs?"),
// not the ring's HBM or fabric traffic.   Build: hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/cu_hog.hip -o tools/_build/libcuhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void cu_hog_kernel(uint64_t ticks, unsigned* sink) {
    extern __shared__ char lds[];
    const uint64_t t0 = wall_clock64();
    if (threadIdx.x == 0) lds[0] = 1;                  // (the allocation must be real)
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (sink && lds[0] == 7 && threadIdx.x == 0) sink[blockIdx.x] = 1;
}

extern "C" int cuhog_launch(int n_cus, double usec, void* stream) {
    static bool attr = false;
    const int lds = 160 * 1024;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cu_hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
        attr = true;
    }
    if (n_cus <= 0) return 0;
    const uint64_t ticks = (uint64_t)(usec * 100.0);   // wall_clock64(): 100 MHz
    hipLaunchKernelGGL(cu_hog_kernel, dim3((unsigned)n_cus), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), ticks, (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
