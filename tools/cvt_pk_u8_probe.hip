// cvt_pk_u8_probe -- what v_cvt_pk_u8_f32 does with fractions, negatives, overflow, NaN (dev tool, round 6: the ME_GG8 pack in gemm3.hip relies on it).
//   hipcc -O2 --offload-arch=gfx950 tools/cvt_pk_u8_probe.hip -o tools/_build/cvt_pk_u8_probe && tools/_build/cvt_pk_u8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}
int main() {
    float h[] = {-5.f, -0.6f, -0.5f, -0.4f, 0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.6f, 255.f, 255.5f, 256.f, 300.f, 1e9f, __builtin_nanf(""), __builtin_inff()};
    int n = sizeof(h) / 4;
    float* d; unsigned* o; unsigned r[32];
    hipMalloc(&d, 128); hipMalloc(&o, 128);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%g -> %08x (byte %u)\n", h[i], r[i], (r[i] >> 8) & 255);
    return 0;
}
