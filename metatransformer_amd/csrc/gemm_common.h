// gemm_common.h -- parameter block and fused epilogue shared by the GEMM kernel families.
#pragma once
#include "common.h"

struct GemmParams {
    const void* A; const void* B; void* C;
    int64_t M, N, K, lda, ldb, ldc;
    int c_dtype, act;
    float alpha, beta;
    const float* bias; const float* colscale;
    void* preact; int64_t ldpre; int preact_dtype;
    const void* aux; int64_t ldaux; int aux_dtype;
    const void* residual; int64_t ldres; int res_dtype;
    int64_t res_row_mod, out_group_rows, out_group_stride, out_row_offset;
    int tiles_m, tiles_n;
    int split_k, ksteps_per_split;          // split-K (wgrad): grid.y = split_k, slab z written to C + z*M*ldc
};


// One accumulator quad: 4 consecutive output columns n..n+3 of output row m (see include/metaenc.h for the order).
__device__ __forceinline__ void epilogue_quad(const GemmParams& p, int64_t m, int64_t n, f32x4 v) {
    v *= p.alpha;
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.preact) store4_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n, v);
    if (p.act == ME_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    }
    if (p.aux) {
        const f32x4 a = load4_as_f32(p.aux, p.aux_dtype, m * p.ldaux + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(a[e]);
    }
    if (p.colscale) v *= *reinterpret_cast<const f32x4*>(p.colscale + n);
    if (p.residual) {
        const int64_t rr = p.res_row_mod ? (m % p.res_row_mod) : m;
        v += load4_as_f32(p.residual, p.res_dtype, rr * p.ldres + n);
    }
    const int64_t orow = p.out_group_rows
                             ? (m / p.out_group_rows) * p.out_group_stride + (m % p.out_group_rows) + p.out_row_offset
                             : m;
    if (p.beta != 0.0f) v += p.beta * load4_as_f32(p.C, p.c_dtype, orow * p.ldc + n);
    store4_from_f32(p.C, p.c_dtype, orow * p.ldc + n, v);
}


// kernel families (each in its own translation unit)
int launch_g256(const GemmParams& p, int op, int bn, hipStream_t stream);
bool g256_supported(const GemmParams& p, int op);
