// Developer probe: issue cost (shader clocks per wave64 instruction) of the VALU operations the attention softmax uses, one
// wave and two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o tools/probe_valu tools/probe_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
template <int OP> __global__ void k(long long* out, float* sink, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 16; ++it) {
        if (OP == 0) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 2) { REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 3) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 4) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
        if (OP == 5) { REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
        if (OP == 6) { REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 7) { REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %4, %5\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %6, %7\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 8) { REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3\n v_add_f32 %2, %2, %3\n v_fma_f32 %3, %3, %4, %5\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %6, %7\n v_add_f32 %6, %6, %7\n v_fma_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    }
    long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1];
}
template <int OP> void run(const char* name) {
    long long* out; float* sink; hipMalloc(&out, 1024 * 16 * 8); hipMalloc(&sink, 1024 * 1024 * 4);
    for (int waves = 4; waves <= 16; waves += 4) {       // 4 waves = one per SIMD, 8 = two per SIMD, ...
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, out, sink, 0.001f);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, out, sink, 0.001f);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * 16); hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += h[b * 16 + w];
        printf("%-44s %d waves/SIMD: %6.2f clocks per instruction per wave = %5.2f per instruction per SIMD\n", name, waves / 4, s / (256.0 * waves) / (16 * 64), s / (256.0 * waves) / (16 * 64) / (waves / 4));
    }
}
int main() {
    run<0>("v_exp_f32"); run<1>("v_fma_f32"); run<2>("v_max3_f32"); run<3>("v_cvt_pk_bf16_f32"); run<4>("v_pk_fma_f32"); run<5>("v_pk_add_f32");
    run<6>("v_add_f32"); run<7>("v_exp_f32 / v_fma_f32 alternating"); run<8>("1 exp : 2 fma : 1 add");
    return 0;
}
