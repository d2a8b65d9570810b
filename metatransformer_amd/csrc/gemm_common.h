// gemm_common.h -- parameter block and fused epilogue shared by the GEMM kernel families.
#pragma once
#include "common.h"

struct GemmParams {
    const void* A; const void* B; void* C;
    int64_t M, N, K, lda, ldb, ldc;
    int c_dtype, act;
    int flags;                              // ME_GEMM_SAVE_GELU_GRAD / ME_GEMM_AUX_IS_FACTOR
    float alpha, beta;
    const float* bias; const float* colscale;
    void* preact; int64_t ldpre; int preact_dtype;
    const void* aux; int64_t ldaux; int aux_dtype;
    const void* residual; int64_t ldres; int res_dtype;
    int64_t res_row_mod, out_group_rows, out_group_stride, out_row_offset;
    int tiles_m, tiles_n;
    int debug;                              // dev builds only (GemmDev::debug, 0 in the shipped library): 1 = skip the epilogue
    int split_k, ksteps_per_split;          // split-K (wgrad): grid.y = split_k, slab z written to C + z*M*ldc
    float* colsum_ws;                       // TN only: partial column sums of A, [split_k * tiles_n][M] (null = off)
    // g3 tail split (gemm3.hip): tiles [g3_full_tiles, tiles) run as g3_split workgroups each, g3_ktp K-tiles (of 64) per
    // part, raw fp32 partial sums into g3_slabs [g3_split][rows past the full tiles][N]; 0 = every tile is a full tile
    int g3_full_tiles, g3_split, g3_ktp;
    float* g3_slabs;
    int g3_half;                            // resident kernel: tiles [g3_full_tiles, tiles) run as two 128-row items each (real epilogue, no slabs)
    int g3_colgroups;                       // resident kernel: the XCDs split the column tiles into this many groups (1 = every XCD walks whole tile rows)
    unsigned* g3_tickets;                   // resident g3 kernel: per-XCD work counters (16 words apart), null = static schedule
    int64_t slab_stride;                    // g3 wgrad: floats between the split-K slabs in C (>= M * N, padded: see g3_tn_slab_stride)
    const float* row_affine;                // folded LayerNorm (me_gemm_desc.row_affine): [M][2] = (rstd, -rstd * mean), or null
    int row_nparts; float row_eps;          // row_nparts > 0 (me_gemm_desc.row_parts): row_affine holds [row_nparts][M] partial (mean, M2) pairs instead --
                                            // resident kernel only (g3_takes_row_parts); the pairs are LayerNorm(K = 64 row_nparts, row_eps)'s
    const float* col_shift;                 // ... and s[n] = sum_k W'[n, k]
    float* tn_colsum_out;                   // g3 wgrad with the in-kernel fold: where the folded column sums of A go (follows C's beta), or null
    int sk_wgs, sk_upt, sk_levels, sk_l1;   // g3 wgrad on sk_wgs > 0 workgroups that are NOT a multiple of the tile count (gemm3.hip: gemm_g3tn_sk_kernel): sk_levels
                                            // whole split levels of sk_l1 K-tile pairs per tile (one workgroup each, split-major as the uniform grid) + the rest of every
                                            // tile's sk_upt pairs shared evenly, across tile boundaries, by the sk_wgs - sk_levels x tiles workgroups left over
    int a_wrap_kt;                          // one-tile g3 NT kernel: A's K-tile index wraps back to 0 from this K-tile on (me_gemm_desc.a_wrap_k / 64; 0 = off)
    float* row_stats;                       // resident EPI 2 kernel only (me_gemm_desc.row_stats): per-row partial statistics of the OUTPUT,
                                            // [N / 64][M] pairs (mean, M2) over 64-column groups, or null
};


// ---- a split-K weight gradient on W workgroups, W not a multiple of the T tiles (GemmParams::sk_*): S = W / T whole split levels of L1 pairs
// per tile, then the LEFTOVER Ul = U - S L1 pairs of every tile as a tile-major list of T Ul units that the E = W - S T extra workgroups share
// evenly: extra workgroup e owns units [e T Ul / E, (e + 1) T Ul / E) -- it may end one tile's leftover and begin the next.  sk_owner(u) = the
// extra workgroup that holds leftover unit u; a tile's leftover slabs are numbered in K order behind its S level slabs.  Integer arithmetic
// only, the same on host and device.
__host__ __device__ static inline int sk_owner(int64_t u, int E, int64_t TUl) { return (int)(((u + 1) * E - 1) / TUl); }
__host__ __device__ static inline int sk_first(int tile, int E, int Ul, int64_t TUl) { return sk_owner((int64_t)tile * Ul, E, TUl); }
__host__ __device__ static inline int sk_left_parts(int tile, int E, int Ul, int64_t TUl) {
    return Ul > 0 ? sk_owner((int64_t)(tile + 1) * Ul - 1, E, TUl) - sk_first(tile, E, Ul, TUl) + 1 : 0;
}
// slabs of tile `tile` under GemmParams p (p.sk_wgs > 0)
#define ME_SK_PARTS(p, tile) ((p).sk_levels + sk_left_parts((tile), (p).sk_wgs - (p).sk_levels * (p).tiles_m * (p).tiles_n, (p).sk_upt - (p).sk_levels * (p).sk_l1, \
                                                            (int64_t)(p).tiles_m * (p).tiles_n * ((p).sk_upt - (p).sk_levels * (p).sk_l1)))

// One accumulator quad: 4 consecutive output columns n..n+3 of output row m (see include/metaenc.h for the order).
__device__ __forceinline__ void epilogue_quad(const GemmParams& p, int64_t m, int64_t n, f32x4 v) {
    v *= p.alpha;
    if (p.row_affine) v = v * p.row_affine[2 * m] + *reinterpret_cast<const f32x4*>(p.col_shift + n) * p.row_affine[2 * m + 1];
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.flags & ME_GEMM_SAVE_GELU_GRAD) {        // (act = GELU and preact set: fill_params) -- GELU and its derivative from one Phi / Gaussian
        f32x4 dv;
        gelu_erf_pair4(v, v, dv);
        store4_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n, dv);
    } else {
        if (p.preact) store4_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n, v);
        if (p.act == ME_ACT_GELU) v = gelu_for4(v, p.c_dtype);
    }
    if (p.aux) {
        const f32x4 a = load4_as_f32(p.aux, p.aux_dtype, m * p.ldaux + n);
        v *= (p.flags & ME_GEMM_AUX_IS_FACTOR) ? a : gelu_grad_for4(a, p.c_dtype);
    }
    if (p.colscale) v *= *reinterpret_cast<const f32x4*>(p.colscale + n);
    if (p.residual) {
        const int64_t rr = p.res_row_mod ? (m % p.res_row_mod) : m;
        v += load4_as_f32(p.residual, p.res_dtype, rr * p.ldres + n);
    }
    const int64_t orow = p.out_group_rows
                             ? (m / p.out_group_rows) * p.out_group_stride + (m % p.out_group_rows) + p.out_row_offset
                             : m;
    if (p.c_dtype == ME_BF16X3) {                      // fp32 result as three bf16 planes [hi | lo | hi] (the next Linear's A operand)
        store4_split3(reinterpret_cast<uint16_t*>(p.C) + orow * p.ldc, p.N, n, v);
        return;
    }
    if (p.c_dtype == ME_BF16X2) {                      // ... as two planes [hi | lo]
        store4_split2(reinterpret_cast<uint16_t*>(p.C) + orow * p.ldc, p.N, n, v);
        return;
    }
    if (p.beta != 0.0f) v += p.beta * load4_as_f32(p.C, p.c_dtype, orow * p.ldc + n);
    store4_from_f32(p.C, p.c_dtype, orow * p.ldc + n, v);
}



// The same without the GELU forms (no act / preact / aux): what the split-K folds of the weight gradients and of the
// residual GEMMs need.  A separate function so that those fold kernels do not carry the registers of the erf arithmetic
// (they are bandwidth-bound: occupancy is their throughput).
__device__ __forceinline__ void epilogue_quad_lin(const GemmParams& p, int64_t m, int64_t n, f32x4 v) {
    v *= p.alpha;
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.colscale) v *= *reinterpret_cast<const f32x4*>(p.colscale + n);
    if (p.residual) {
        const int64_t rr = p.res_row_mod ? (m % p.res_row_mod) : m;
        v += load4_as_f32(p.residual, p.res_dtype, rr * p.ldres + n);
    }
    const int64_t orow = p.out_group_rows
                             ? (m / p.out_group_rows) * p.out_group_stride + (m % p.out_group_rows) + p.out_row_offset
                             : m;
    if (p.c_dtype == ME_BF16X3) {
        store4_split3(reinterpret_cast<uint16_t*>(p.C) + orow * p.ldc, p.N, n, v);
        return;
    }
    if (p.c_dtype == ME_BF16X2) {
        store4_split2(reinterpret_cast<uint16_t*>(p.C) + orow * p.ldc, p.N, n, v);
        return;
    }
    if (p.beta != 0.0f) v += p.beta * load4_as_f32(p.C, p.c_dtype, orow * p.ldc + n);
    store4_from_f32(p.C, p.c_dtype, orow * p.ldc + n, v);
}

// Eight consecutive output columns n..n+7 of output row m (two quads), with 16-byte accesses on the bf16 side.
__device__ __forceinline__ void load8_as_f32(const void* base, int dt, int64_t idx, f32x4& a, f32x4& b) {
    if (dt == ME_BF16) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(base) + idx);
        a[0] = __uint_as_float(raw[0] << 16); a[1] = __uint_as_float(raw[0] & 0xffff0000u);
        a[2] = __uint_as_float(raw[1] << 16); a[3] = __uint_as_float(raw[1] & 0xffff0000u);
        b[0] = __uint_as_float(raw[2] << 16); b[1] = __uint_as_float(raw[2] & 0xffff0000u);
        b[2] = __uint_as_float(raw[3] << 16); b[3] = __uint_as_float(raw[3] & 0xffff0000u);
    } else {
        a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
        b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx + 4);
    }
}
__device__ __forceinline__ void store8_from_f32(void* base, int dt, int64_t idx, f32x4 a, f32x4 b) {
    if (dt == ME_BF16) {
        bf16x8 o;
        o[0] = (bf16_t)a[0]; o[1] = (bf16_t)a[1]; o[2] = (bf16_t)a[2]; o[3] = (bf16_t)a[3];
        o[4] = (bf16_t)b[0]; o[5] = (bf16_t)b[1]; o[6] = (bf16_t)b[2]; o[7] = (bf16_t)b[3];
        *reinterpret_cast<bf16x8*>(reinterpret_cast<uint16_t*>(base) + idx) = o;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = a;
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx + 4) = b;
    }
}
__device__ __forceinline__ void epilogue_oct(const GemmParams& p, int64_t m, int64_t n, f32x4 v0, f32x4 v1) {
    v0 *= p.alpha; v1 *= p.alpha;
    if (p.row_affine) {
        const float ra = p.row_affine[2 * m], rb = p.row_affine[2 * m + 1];
        v0 = v0 * ra + *reinterpret_cast<const f32x4*>(p.col_shift + n) * rb;
        v1 = v1 * ra + *reinterpret_cast<const f32x4*>(p.col_shift + n + 4) * rb;
    }
    if (p.bias) {
        v0 += *reinterpret_cast<const f32x4*>(p.bias + n);
        v1 += *reinterpret_cast<const f32x4*>(p.bias + n + 4);
    }
    if (p.flags & ME_GEMM_SAVE_GELU_GRAD) {        // (act = GELU and preact set: fill_params) -- the pair from one Phi / Gaussian: erf form, as the resident kernel
        f32x4 d0, d1;
        gelu_erf_pair4(v0, v0, d0);
        gelu_erf_pair4(v1, v1, d1);
        store8_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n, d0, d1);
    } else {
        if (p.preact) store8_from_f32(p.preact, p.preact_dtype, m * p.ldpre + n, v0, v1);
        if (p.act == ME_ACT_GELU) { v0 = gelu_for4(v0, p.c_dtype); v1 = gelu_for4(v1, p.c_dtype); }
    }
    if (p.aux) {
        f32x4 a0, a1;
        load8_as_f32(p.aux, p.aux_dtype, m * p.ldaux + n, a0, a1);
        if (p.flags & ME_GEMM_AUX_IS_FACTOR) { v0 *= a0; v1 *= a1; }
        else { v0 *= gelu_grad_for4(a0, p.c_dtype); v1 *= gelu_grad_for4(a1, p.c_dtype); }
    }
    if (p.colscale) {
        v0 *= *reinterpret_cast<const f32x4*>(p.colscale + n);
        v1 *= *reinterpret_cast<const f32x4*>(p.colscale + n + 4);
    }
    if (p.residual) {
        const int64_t rr = p.res_row_mod ? (m % p.res_row_mod) : m;
        f32x4 r0, r1;
        load8_as_f32(p.residual, p.res_dtype, rr * p.ldres + n, r0, r1);
        v0 += r0; v1 += r1;
    }
    const int64_t orow = p.out_group_rows
                             ? (m / p.out_group_rows) * p.out_group_stride + (m % p.out_group_rows) + p.out_row_offset
                             : m;
    if (p.c_dtype == ME_BF16X3) {
        uint16_t* row = reinterpret_cast<uint16_t*>(p.C) + orow * p.ldc;
        store4_split3(row, p.N, n, v0);
        store4_split3(row, p.N, n + 4, v1);
        return;
    }
    if (p.c_dtype == ME_BF16X2) {
        uint16_t* row = reinterpret_cast<uint16_t*>(p.C) + orow * p.ldc;
        store4_split2(row, p.N, n, v0);
        store4_split2(row, p.N, n + 4, v1);
        return;
    }
    if (p.beta != 0.0f) {
        f32x4 c0, c1;
        load8_as_f32(p.C, p.c_dtype, orow * p.ldc + n, c0, c1);
        v0 += p.beta * c0; v1 += p.beta * c1;
    }
    store8_from_f32(p.C, p.c_dtype, orow * p.ldc + n, v0, v1);
}

// which specialised epilogue covers this call:
//   0 alpha*acc + bias (* colscale)   1 ... -> (store pre-activation) -> GELU   2 ... + bf16 residual row operand
//   3 ... * gelu'(bf16 aux row operand)   4 generic (anything include/metaenc.h allows)   5 raw fp32 split-K slab
static inline int pick_epi(const GemmParams& p) {
    if (p.split_k > 1) return 5;
    if (me_is_planes(p.c_dtype)) return 4;  // (plane stores live in the generic epilogue only)
    if (p.flags || p.row_affine) return 4;  // (the resident g3 kernel has its own forms of these: launch_g3)
    if (p.beta != 0.0f || p.out_group_rows != 0 || p.res_row_mod != 0) return 4;
    const int nrow = (p.residual ? 1 : 0) + (p.aux ? 1 : 0);
    if (nrow > 1) return 4;
    if (p.act == ME_ACT_GELU) return (nrow == 0) ? 1 : 4;
    if (p.preact) return 4;
    if (p.residual) return p.res_dtype == ME_BF16 ? 2 : 4;
    if (p.aux) return p.aux_dtype == ME_BF16 ? 3 : 4;
    return 0;
}

// ... and the two halves of the "saved gelu'" pair of the training MLP, which pick_epi sends to the generic epilogue (4) because
// they carry flags: 6 = acc * bf16 aux row operand (ME_GEMM_AUX_IS_FACTOR: the fc2 dgrad), 7 = GELU with gelu'(h) saved as the
// pre-activation (ME_GEMM_SAVE_GELU_GRAD: fc1 forward in training).  The one-tile-per-workgroup kernels (g2b, g3) have straight-line
// forms of both: at the reference's batch sizes (M = 3 072 .. 8 224) the generic epilogue cost 45 us against 29 us per launch.
static inline int pick_epi_ex(const GemmParams& p) {
    const int e = pick_epi(p);
    // three-plane (ME_BF16X3) outputs of the MLP of an ME_BF16X3 Block: straight-line forms in the one-tile g3 kernel (9: bias + erf GELU,
    // optionally saving gelu' as an fp32 pre-activation; 10: x fp32 row operand); the other families take the generic epilogue for them
    if (me_is_planes(p.c_dtype) && p.split_k <= 1 && !p.row_affine && !p.colscale && !p.residual && p.beta == 0.0f && p.out_group_rows == 0 &&
        p.res_row_mod == 0 && p.ldc % 8 == 0 && p.N % 8 == 0) {
        if (p.act == ME_ACT_GELU && !p.aux && (!p.preact ? p.flags == 0 : (p.flags == ME_GEMM_SAVE_GELU_GRAD && p.preact_dtype == ME_F32 && p.ldpre % 4 == 0)))
            return 9;
        if (p.act == ME_ACT_NONE && p.aux && p.aux_dtype == ME_F32 && p.flags == ME_GEMM_AUX_IS_FACTOR && !p.preact && p.ldaux % 4 == 0) return 10;
    }
    // bias + fp32 residual -> fp32 (8): the residual GEMMs of an fp32 token stream (autocast recipes, ME_BF16X3 Blocks)
    if (e == 4 && p.c_dtype == ME_F32 && p.residual && p.res_dtype == ME_F32 && !p.aux && !p.preact && p.act == ME_ACT_NONE && !p.flags &&
        !p.row_affine && p.beta == 0.0f && p.out_group_rows == 0 && p.res_row_mod == 0 && p.split_k <= 1 && p.ldres % 4 == 0)
        return 8;
    if (e != 4 || me_is_planes(p.c_dtype) || p.row_affine || p.colscale || p.residual || p.beta != 0.0f || p.out_group_rows != 0 || p.res_row_mod != 0) return e;
    if (p.flags == ME_GEMM_AUX_IS_FACTOR && p.act == ME_ACT_NONE && p.aux && (p.aux_dtype == ME_BF16 || p.aux_dtype == ME_GG8) && !p.preact) return 6;
    if (p.flags == ME_GEMM_SAVE_GELU_GRAD && p.act == ME_ACT_GELU && p.preact && !p.aux) return 7;
    return e;
}

// kernel families (each in its own translation unit)
int launch_g2b(const GemmParams& p, int op, int bm, int bn, hipStream_t stream);
bool g2b_supported(const GemmParams& p, int op);
int launch_g3(const GemmParams& p, int epi, void* ws, hipStream_t stream);      // ws = nullptr: one tile per workgroup
bool g3_supported(const GemmParams& p, int op);
bool g3_emits_row_stats(const GemmParams& p);          // will launch_g3 run the resident residual kernel that can emit p.row_stats?
bool g3_takes_row_parts(const GemmParams& p);          // ... the resident folded-LayerNorm epilogue that consumes such partials directly (p.row_nparts)?
bool g3_takes_gg8(const GemmParams& p);         // ME_GG8 preact / aux: the resident kernel's PRE 6 forms only
size_t g3_workspace_bytes();
int launch_g3_tn(const GemmParams& p, hipStream_t stream);      // p.split_k slabs into p.C, p.ksteps_per_split K-tiles of 64 each
int launch_g3_tn_sk(const GemmParams& p, hipStream_t stream);   // the balanced static partition (p.sk_wgs workgroups; p.split_k = the most slabs a tile gets)
// A/B arm, measured and NOT shipped (round 4): -DG3_TN_FOLD=1 = the split-K fold of the wgrad kernel INSIDE its launch.  Same-box
// result: wgrad 211 us per launch against 176 us for kernel + separate fold launch (train step 32.8 vs 31.35 ms): the S partial
// tiles of a tile are 64 MB per launch either way, and inside the launch their write-through, the wait for the tile row and the
// read-back are exposed one after the other on every CU, where the separate fold streams them at full-chip bandwidth while the
// next kernel's launch overlaps the tail (the guide's "splitk-seam" verdict, reproduced at this size).
#ifndef G3_TN_FOLD
#define G3_TN_FOLD 0
#endif
// the same with the split-K fold inside the launch (p.C = the real output, p.g3_slabs = p.split_k slabs, p.g3_tickets = one zeroed
// counter per tile row); only when every workgroup of the launch is resident at once (g3_tn_fold_ok)
int launch_g3_tn_fold(const GemmParams& p, hipStream_t stream);
bool g3_tn_fold_ok(const GemmParams& p, int split_k);
bool g3_tn_supported(const GemmParams& p);
// floats between the split-K slabs of the g3 wgrad kernel.  (Padding the stride off the 256 KiB multiples the encoder's
// shapes give was tried against HBM channel aliasing in the fold: 34 us against 23, i.e. worse -- the slabs stay dense.)
static inline int64_t g3_tn_slab_stride(int64_t M, int64_t N) { return M * N; }

// Dev switches for A/B runs (tools/gemm_dev): they exist only in the dev build of the library (-DME_DEV, built by
// `python -m metatransformer_amd.build --dev` into tools/_build/); the shipped library has no knobs and reads no
// environment variables.
struct GemmDev { int family, bn, debug, tail_split, g3_persistent; };
#ifdef ME_DEV
constexpr bool kMeDev = true;
#define ME_DEV_ONLY(...) __VA_ARGS__
int launch_g3p(const GemmParams& p, int epi, void* ws, hipStream_t stream);      // gemm3_dev.hip
extern GemmDev g_gemm_dev;
extern void* g_gemm_dev_trace;      // device buffer for the resident kernel's time stamps (tools/gemm_dev --trace)
static inline GemmDev gemm_dev() { return g_gemm_dev; }
#else
constexpr bool kMeDev = false;
#define ME_DEV_ONLY(...)
static inline int launch_g3p(const GemmParams&, int, void*, hipStream_t) { return ME_ERR_UNSUPPORTED; }
static inline GemmDev gemm_dev() { return GemmDev{-1, 0, 0, 1, 1}; }
#endif
