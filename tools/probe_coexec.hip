// Developer probe: do MFMA and VALU work co-execute on one SIMD (a) from two different waves, (b) interleaved in one wave?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_coexec tools/probe_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// MODE: what a wave does.  role = (wave index / 4) & 1 decides between the two programs when MIX.
//  0: all waves MFMA16 (64 per iteration, 4 independent accumulators)    1: all waves VALU (256 v_fma per iteration)
//  2: even SIMD-mates MFMA16, odd VALU                                   3: one stream: MFMA16 + 4 v_fma, repeated
//  4: all waves MFMA32 (32 per iteration)                                5: one stream: MFMA32 + 8 v_fma
//  6: MFMA16 waves vs v_exp waves                                        7: one stream: MFMA16 + 2 v_exp
template <int MODE> __global__ void k(long long* out, float* sink, float seed) {
    const int wave = threadIdx.x >> 6;
    const int role = (wave >> 2) & 1;
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    bf16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(seed * e); fb[e] = (__bf16)(seed + e); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0, d1;
    for (int e = 0; e < 16; ++e) { d0[e] = 0; d1[e] = 0; }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#define VFMA4 asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
#define VEXP2 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a6), "+v"(a7));
#define M16(C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, C, 0, 0, 0);
#define M32(D) D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, D, 0, 0, 0);
    for (int it = 0; it < 16; ++it) {
        const bool mf = MODE == 0 || MODE == 4 || ((MODE == 2 || MODE == 6) && role == 0);
        const bool va = MODE == 1 || (MODE == 2 && role == 1);
        const bool ve = MODE == 6 && role == 1;
        if (MODE == 0 || ((MODE == 2 || MODE == 6) && role == 0)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { M16(c0) M16(c1) M16(c2) M16(c3) }
        }
        if (va) {
#pragma unroll
            for (int r = 0; r < 64; ++r) { VFMA4 }
        }
        if (ve) {
#pragma unroll
            for (int r = 0; r < 64; ++r) { VEXP2 }
        }
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { M16(c0) VFMA4 M16(c1) VFMA4 M16(c2) VFMA4 M16(c3) VFMA4 }
        }
        if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { M32(d0) M32(d1) }
        }
        if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { M32(d0) VFMA4 VFMA4 M32(d1) VFMA4 VFMA4 }
        }
        if (MODE == 7) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { M16(c0) VEXP2 M16(c1) VEXP2 M16(c2) VEXP2 M16(c3) VEXP2 }
        }
        (void)mf;
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    float acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name, int waves) {
    long long* out; float* sink; hipMalloc(&out, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    hipMemset(out, 0, 256 * 16 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, sink, 0.001f);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, sink, 0.001f);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 16); hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s0 = 0, s1 = 0; int n0 = 0, n1 = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { if ((w >> 2) & 1) { s1 += h[b * 16 + w]; ++n1; } else { s0 += h[b * 16 + w]; ++n0; } }
    printf("%-58s %2d waves/CU: first-role waves %8.0f clocks", name, waves, n0 ? s0 / n0 : 0.0);
    if (n1) printf("   second-role waves %8.0f clocks", s1 / n1);
    printf("\n");
    hipFree(out); hipFree(sink);
}
int main() {
    run<0>("MFMA16 only (1024 per wave)", 4); run<0>("MFMA16 only (1024 per wave)", 8);
    run<1>("v_fma only (4096 per wave)", 4); run<1>("v_fma only (4096 per wave)", 8);
    run<2>("SIMD mates: MFMA16 wave + v_fma wave", 8); run<2>("two of each per SIMD", 16);
    run<3>("one stream, MFMA16 + 4 v_fma (1024 + 4096)", 4); run<3>("one stream, MFMA16 + 4 v_fma", 8);
    run<4>("MFMA32 only (512 per wave)", 4);
    run<5>("one stream, MFMA32 + 8 v_fma (512 + 4096)", 4); run<5>("one stream, MFMA32 + 8 v_fma", 8);
    run<6>("SIMD mates: MFMA16 wave + v_exp wave (2048 exp)", 8); run<6>("two of each per SIMD", 16);
    run<7>("one stream, MFMA16 + 2 v_exp (1024 + 2048)", 4); run<7>("one stream, MFMA16 + 2 v_exp", 8); run<7>("one stream, MFMA16 + 2 v_exp", 16);
    return 0;
}
