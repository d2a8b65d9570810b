// common.h -- shared device/host helpers for libmetaenc (gfx950 / CDNA4 only).
#pragma once
// Compile-time cache-policy arms (A/B builds only; all default to the plain policy): non-temporal loads / stores for streams that
// are written or read exactly once (tools/r4_policy_builds.sh).  __builtin_nontemporal_* lower to global_load / global_store ... nt.
#ifndef ME_POL_LN_ST
#define ME_POL_LN_ST 0
#endif
#ifndef ME_POL_LN_LD
#define ME_POL_LN_LD 0
#endif
#ifndef ME_POL_ATTN_ST
#define ME_POL_ATTN_ST 0
#endif
#ifndef ME_POL_SLAB
#define ME_POL_SLAB 0
#endif
#ifndef ME_POL_ADAMW
#define ME_POL_ADAMW 0
#endif
#define ME_NT_LOAD(POL, ptr) ((POL) ? __builtin_nontemporal_load(ptr) : *(ptr))
#define ME_NT_STORE(POL, val, ptr) do { if (POL) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/metaenc.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
// ME_GG8 (include/metaenc.h): gelu'(h) in eight bits, g = ME_GG8_LO + ME_GG8_STEP * q
constexpr float ME_GG8_LO = -0.13f, ME_GG8_STEP = 1.26f / 255.0f;
constexpr int ME_STATS_GROUP = 256;      // columns per partial (mean, M2) pair of me_gemm_desc.row_stats / row_parts: one output tile's width
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define ME_WAVE 64

// ---- host-side error plumbing (api.cpp owns the storage)
void me_set_error(const char* fmt, ...);
#define ME_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            me_set_error(__VA_ARGS__);          \
            return ME_ERR_ARG;                  \
        }                                       \
    } while (0)
#define ME_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            me_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return ME_ERR_HIP;                                                       \
        }                                                                            \
    } while (0)

// hipFuncSetAttribute (dynamic LDS size) is per DEVICE: a per-kernel `static bool` would leave the second GPU of a
// single-process multi-GPU host (nn.DataParallel, Audio/src/traintest.py) without it.  Benign if two threads race.
struct OncePerDevice {
    bool done[64] = {};
    bool need() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

// optional per-launch timing (me_gemm_profile_enable / _read, api.hip): brackets the launches an entry point makes with
// HIP events on its stream while profiling is on; free when off.  op: ME_GEMM_NT / ME_GEMM_TN or an ME_PROF_* code.
struct ProfScope {
    hipEvent_t e0 = nullptr;
    hipStream_t stream;
    int op, dt;
    int64_t M, N, K;
    int plan = 0;               // (GEMM records: me_gemm_profile_rec.plan, filled in by the planner)
    ProfScope(int op, int dt, int64_t M, int64_t N, int64_t K, hipStream_t stream);
    ~ProfScope();
};

static inline size_t me_dtype_size(int dt) { return dt == ME_F32 ? 4 : dt == ME_GG8 ? 1 : 2; }
static inline bool me_dtype_ok(int dt) { return dt == ME_F32 || dt == ME_BF16; }                    // compute dtypes
static inline bool me_storage_dtype_ok(int dt) { return me_dtype_ok(dt) || dt == ME_F16; }          // me_cast only
static inline bool me_out_dtype_ok(int dt) { return me_dtype_ok(dt) || dt == ME_BF16X3; }           // me_layernorm_fwd y (and me_gemm C, which also takes ME_BF16X2)
static inline bool me_is_planes(int dt) { return dt == ME_BF16X3 || dt == ME_BF16X2; }             // fp32 values stored as bf16 hi / lo planes

// ---- LayerNorm backward with the dgamma / dbeta fold deferred (layernorm.hip; me_block_bwd folds both LayerNorms of a block in one launch)
constexpr int LN_FOLD_SETS = 4;
struct me_ln_fold_set { const float* partial; int nrows; float* dgamma; float* dbeta; int accumulate; };
struct me_ln_fold_batch { me_ln_fold_set set[LN_FOLD_SETS]; int cols; };
int me_ln_bwd_deferred(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd, const float* gamma,
                       const void* dres, int dres_dtype, void* dx, int dx_dtype, float* dgamma, float* dbeta, int accumulate_affine,
                       int64_t rows, int cols, void* workspace, void* stream, me_ln_fold_set* defer);
int me_ln_bwd_fold_sets(const me_ln_fold_set* sets, int n, int cols, void* stream);      // n <= LN_FOLD_SETS

// ---- device helpers
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t lo16) { return __uint_as_float(lo16 << 16); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 4 consecutive elements at element index idx (idx % 4 == 0) of a [.., ld] matrix of dtype dt -> fp32
__device__ __forceinline__ f32x4 load4_as_f32(const void* base, int dt, int64_t idx) {
    if (dt == ME_BF16) {
        u32x2 raw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(base) + idx);
        f32x4 r;
        r[0] = __uint_as_float(raw[0] << 16);
        r[1] = __uint_as_float(raw[0] & 0xffff0000u);
        r[2] = __uint_as_float(raw[1] << 16);
        r[3] = __uint_as_float(raw[1] & 0xffff0000u);
        return r;
    }
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
}
__device__ __forceinline__ void store4_from_f32(void* base, int dt, int64_t idx, f32x4 v) {
    if (dt == ME_BF16) {
        bf16x4 o;
        o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<uint16_t*>(base) + idx) = o;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
    }
}
// ME_BF16X3 (include/metaenc.h): four consecutive columns c..c+3 of a row of `cols` fp32 values, stored into the row's three bf16
// planes (row = first bf16 of the 3 * cols long row).  Left-operand order [hi | lo | hi], right-operand order [hi | hi | lo].
__device__ __forceinline__ void store4_split3(uint16_t* row, int64_t cols, int64_t c, f32x4 v, bool right_operand = false) {
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (bf16_t)v[e];
        l[e] = (bf16_t)(v[e] - (float)h[e]);          // exact difference (hi carries the top 8 bits of v), then 8 more bits
    }
    *reinterpret_cast<bf16x4*>(row + c) = h;
    *reinterpret_cast<bf16x4*>(row + cols + c) = right_operand ? h : l;
    *reinterpret_cast<bf16x4*>(row + 2 * cols + c) = right_operand ? l : h;
}
// ME_BF16X2: [hi | lo] only
__device__ __forceinline__ void store4_split2(uint16_t* row, int64_t cols, int64_t c, f32x4 v) {
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (bf16_t)v[e];
        l[e] = (bf16_t)(v[e] - (float)h[e]);
    }
    *reinterpret_cast<bf16x4*>(row + c) = h;
    *reinterpret_cast<bf16x4*>(row + cols + c) = l;
}

__device__ __forceinline__ float load1_as_f32(const void* base, int dt, int64_t idx) {
    if (dt == ME_BF16) return bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(base)[idx]);
    return reinterpret_cast<const float*>(base)[idx];
}
__device__ __forceinline__ void store1_from_f32(void* base, int dt, int64_t idx, float v) {
    if (dt == ME_BF16) reinterpret_cast<bf16_t*>(base)[idx] = (bf16_t)v;
    else reinterpret_cast<float*>(base)[idx] = v;
}

// counter-based RNG for the stochastic ops (dropout / drop-path / attention dropout): splitmix64 of (seed, element index)
// -> uniform [0, 1).  The same (seed, index) regenerates the same mask in backward; nothing is stored.
__device__ __forceinline__ float u01_hash(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);      // 24 random bits -> [0, 1)
}

// exact-erf GELU and its derivative (nn.GELU(approximate='none')).
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, branch-free) instead of libm's erff, which costs several times
// more VALU in the GEMM epilogues.  Written on 4-wide vectors so hipcc emits packed fp32 math (v_pk_fma_f32 /
// v_pk_mul_f32: two lanes-elements per instruction); only rcp and exp2 are per element.  Constants are folded so that
//   w = |x| * sqrt(log2(e)/2)            ->  e^{-x^2/2} = 2^{-w^2}          (one v_exp_f32, no extra scale)
//   t = 1 / (1 + p*|x|/sqrt(2))          =   rcp(fma(w, p/sqrt(log2 e), 1))
//   q = 0.5 * poly(t) * 2^{-w^2}         =   1 - Phi(|x|)                    (0.5 folded into the coefficients)
//   Phi(x) = 0.5 + copysign(0.5 - q, x);   gelu = x*Phi ;   gelu' = Phi + x * e^{-x^2/2} / sqrt(2 pi)
// The same exponential serves the Gaussian term of the derivative.
__device__ __forceinline__ void phi_parts4(f32x4 x, f32x4& phi, f32x4& gauss) {
    const f32x4 ax = {fabsf(x[0]), fabsf(x[1]), fabsf(x[2]), fabsf(x[3])};
    const f32x4 w = ax * 0.84932180028801907f;                    // sqrt(log2(e) / 2)
    const f32x4 u = w * 0.27273748087922245f + 1.0f;              // 0.3275911 / sqrt(2) / sqrt(log2(e)/2) ... = p*|x|/sqrt2
    const f32x4 nw2 = -(w * w);
    f32x4 t, e;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t[i] = __builtin_amdgcn_rcpf(u[i]);
        e[i] = __builtin_amdgcn_exp2f(nw2[i]);
    }
    f32x4 poly = t * 0.5307027145f - 0.7265760135f;               // 0.5 * {1.061405429, -1.453152027, ...}
    poly = poly * t + 0.7107068705f;
    poly = poly * t - 0.142248368f;
    poly = poly * t + 0.127414796f;
    const f32x4 q = poly * t * e;                                 // 1 - Phi(|x|)
    const f32x4 hmq = 0.5f - q;
#pragma unroll
    for (int i = 0; i < 4; ++i) phi[i] = 0.5f + copysignf(hmq[i], x[i]);
    gauss = e;
}
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 x) {
    f32x4 phi, g;
    phi_parts4(x, phi, g);
    return x * phi;
}
__device__ __forceinline__ f32x4 gelu_erf_grad4(f32x4 x) {
    f32x4 phi, g;
    phi_parts4(x, phi, g);
    return phi + x * g * 0.3989422804014327f;
}
// both at once (ONE Phi / Gaussian: two separate calls on either side of a branch are not merged by the compiler)
__device__ __forceinline__ void gelu_erf_pair4(f32x4 x, f32x4& y, f32x4& dy) {
    f32x4 phi, g;
    phi_parts4(x, phi, g);
    y = x * phi;
    dy = phi + x * g * 0.3989422804014327f;
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf4(f32x4{x, x, x, x})[0]; }
__device__ __forceinline__ float gelu_erf_grad(float x) { return gelu_erf_grad4(f32x4{x, x, x, x})[0]; }

// ---- bf16-mode GELU: what the epilogues use when the result is ROUNDED TO BF16 (8 significant bits).
// The erf form above spends two quarter-rate transcendentals (v_rcp_f32, v_exp_f32) and ~14 plain VALU operations per element
// on 1.5e-7 accuracy that the bf16 store throws away again; in the GEMM epilogues that arithmetic is what the matrix pipe
// waits for (fc1: 14-23 k of ~50 k clocks per 256 x 256 tile).  Here both functions are written as "the piecewise-linear part
// plus a remainder that is smooth on t = |x| >= 0 and flat beyond T = 4":
//     gelu(x)  = max(x, 0) + r(t),          r(t) = t (Phi(t) - 1)              -> 0    for t -> inf
//     gelu'(x) = 1/2 + copysign(e(t), x),   e(t) = Phi(t) + t phi(t) - 1/2     -> 1/2  for t -> inf
// with r ~ t P(t), e ~ t Q(t) (degree-7 P, Q; t clamped at T): one v_min, seven FMAs and two more operations each, no
// transcendental.  Minimax over the whole real line in the ABSOLUTE error of the result (tools/gelu_poly_fit.py, which also
// evaluates this exact float32 formula): |gelu err| <= 1.9e-4 (2^-12.4), |gelu' err| <= 4.2e-4 -- a fraction of the bf16
// rounding of the stored value for all but the smallest outputs, and P(0) = -1/2, Q(0) = 2 phi(0) are pinned so that
// gelu(x) -> x / 2 and gelu'(x) - 1/2 -> 2 phi(0) x keep their RELATIVE accuracy around zero.  fp32 outputs keep the erf form, and so
// does the training forward that stores gelu AND gelu' (one exponential serves both there: measured no slower than two chains).
// Measured (MI355X, fc1 of config 2): forward-only epilogue 261 -> 236 us per launch, with the pre-activation saved 335 -> 308 us.
// (inline asm for the clamp and the ramp: written as fminf(fabsf(x), 4) / fmaxf(x, 0) hipcc first canonicalises the operand with
// a v_max_f32 x, x each -- two of the ten operations per element.  The NEGATIVE leading coefficients go through an opaque
// register: hipcc rewrites t * (-c) as (-t) * c and then, finding no negation modifier it can use on the packed FMA, negates t
// with a v_xor per element.)
__device__ __forceinline__ f32x4 gelu_clamp4(f32x4 x) {
    f32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_min_f32_e64 %0, |%1|, 4.0" : "=v"(t[i]) : "v"(x[i]));
    return t;
}
__device__ __forceinline__ float gelu_opaque(float c) {
    asm("" : "+v"(c));
    return c;
}
__device__ __forceinline__ f32x4 gelu_bf16_from_t4(f32x4 x, f32x4 t) {
    f32x4 p = t * gelu_opaque(-7.715494112e-06f) + 7.679707389e-04f;
    p = p * t - 1.074723536e-02f;
    p = p * t + 6.040871459e-02f;
    p = p * t - 1.453980336e-01f;
    p = p * t + 4.842520777e-02f;
    p = p * t + 3.880065109e-01f;
    p = p * t - 0.5f;
    f32x4 relu;
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_max_f32_e32 %0, 0, %1" : "=v"(relu[i]) : "v"(x[i]));
    return t * p + relu;
}
__device__ __forceinline__ f32x4 gelu_bf16_grad_from_t4(f32x4 x, f32x4 t) {
    f32x4 q = t * gelu_opaque(-5.104965709e-04f) + 7.796528677e-03f;
    q = q * t - 4.375422062e-02f;
    q = q * t + 9.542723363e-02f;
    q = q * t + 1.391019310e-02f;
    q = q * t - 3.006181013e-01f;
    q = q * t + 1.278037861e-02f;
    q = q * t + 7.978845608e-01f;
    const f32x4 e = t * q;
    return f32x4{0.5f + copysignf(e[0], x[0]), 0.5f + copysignf(e[1], x[1]), 0.5f + copysignf(e[2], x[2]), 0.5f + copysignf(e[3], x[3])};
}
__device__ __forceinline__ f32x4 gelu_bf16_4(f32x4 x) { return gelu_bf16_from_t4(x, gelu_clamp4(x)); }
__device__ __forceinline__ f32x4 gelu_bf16_grad4(f32x4 x) { return gelu_bf16_grad_from_t4(x, gelu_clamp4(x)); }
// form by the dtype the result is stored in (wave-uniform)
__device__ __forceinline__ f32x4 gelu_for4(f32x4 x, int out_dtype) { return out_dtype == ME_BF16 ? gelu_bf16_4(x) : gelu_erf4(x); }
__device__ __forceinline__ f32x4 gelu_grad_for4(f32x4 x, int out_dtype) { return out_dtype == ME_BF16 ? gelu_bf16_grad4(x) : gelu_erf_grad4(x); }

// ---- MFMA "16-byte chunk" abstraction.
// A chunk is 16 bytes of an operand row along the reduction dimension: 8 bf16 or 4 fp32.
// For the 32x32 MFMA shapes lane l supplies, for output row/col (l & 31), the reduction slice owned by
// half h = l >> 5.  mma_chunk() contracts one chunk-pair (both halves) into a 32x32 fp32 accumulator:
//   bf16: one v_mfma_f32_32x32x16_bf16   (16 reduction elements: 8 per half)
//   fp32: four v_mfma_f32_32x32x2_f32    ( 8 reduction elements: 4 per half; exact fp32 fma chain)
// Any assignment of reduction indices to (half, element) is valid as long as A and B use the same one.
// Accumulator layout (both dtypes): acc[r] = D[i = (r&3) + 8*(r>>2) + 4*h][j = l & 31].
template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> {
    typedef bf16x8 type;
    static constexpr int E = 8;   // elements per chunk
};
template <> struct Chunk<float> {
    typedef f32x4 type;
    static constexpr int E = 4;
};

__device__ __forceinline__ f32x16 mma_chunk(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma_chunk(f32x4 a, f32x4 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    return c;
}
// row index inside a 32x32 accumulator tile for register r of half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// per-stream work counters for persistent kernels that CLAIM their items (gemm3.hip): zero between launches, null under hipGraph capture
unsigned* me_work_counters(hipStream_t stream);
