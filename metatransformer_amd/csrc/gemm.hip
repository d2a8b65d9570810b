// gemm.hip -- MFMA GEMM with fused epilogue for the encoder's Linear layers (forward, dgrad, wgrad).
//
// Replaces nn.Linear / GELU / residual-add of the reference Block
// (PointCloud/openpoints/models/layers/attention.py:28,36,56-57; mlp.py:30-34) -- see include/metaenc.h.
//
// Kernel family "g128": 128x128 output tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile made
// of 2x2 MFMA 32x32 accumulators.  Operand rows are staged through LDS as 128-byte rows of reduction data
// (64 bf16 or 32 fp32 per K-step), XOR-swizzled so that the 16-byte fragment reads (row = lane) are
// bank-conflict free, double-buffered with one barrier per K-step, global loads for step t+1 in flight
// while step t computes.  The MFMA is issued with the *weight* tile as the A operand and the *activation*
// tile as the B operand, so every lane owns one output row and 4 consecutive output columns per
// accumulator quad -> 8/16-byte epilogue loads and stores.
//
//   NT: C[M,N] = A[M,K] B[N,K]^T   rows of both operands are K-contiguous: 16-byte chunk copies.
//   TN: C[M,N] = A[K,M]^T B[K,N]   (wgrad) the loader transposes while staging: pairs of reduction rows are
//                                  interleaved into dwords so that LDS rows are again reduction-contiguous.
#include "gemm_common.h"
#include <string.h>
#include <algorithm>
#include <atomic>

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int ROWB = 128;                 // bytes of reduction data per LDS row per K-step
constexpr int TILE_BYTES = BM * ROWB;     // 16 KiB per operand tile
constexpr int NT = 256;

__device__ __forceinline__ int lds_off(int row, int byteoff) {
    const int chunk = byteoff >> 4;
    return row * ROWB + (((chunk ^ ((row >> 1) & 7)) << 4) | (byteoff & 15));
}

struct Stage { u32x4 v[4]; };

// ---- NT loader: operand S[nrows, K] (K contiguous); tile rows r0.., reduction k0..
template <typename T>
__device__ __forceinline__ void load_nt(Stage& s, const T* __restrict__ S, int64_t ld, int64_t nrows, int64_t K,
                                        int64_t r0, int64_t k0, int tid) {
    constexpr int E = 16 / sizeof(T);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int item = tid + NT * i;
        const int chunk = item & 7, row = item >> 3;
        int64_t gr = r0 + row;
        gr = gr < nrows ? gr : nrows - 1;
        const int64_t gk = k0 + chunk * E;
        u32x4 z = {0u, 0u, 0u, 0u};
        s.v[i] = (gk < K) ? *reinterpret_cast<const u32x4*>(S + gr * ld + gk) : z;
    }
}
__device__ __forceinline__ void store_nt(const Stage& s, char* lds, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int item = tid + NT * i;
        const int chunk = item & 7, row = item >> 3;
        *reinterpret_cast<u32x4*>(lds + lds_off(row, chunk * 16)) = s.v[i];
    }
}

// ---- TN loader: operand S[Tred, ncols] (reduction index is the ROW); tile cols c0.., reduction t0..
template <typename T> struct TnLoader;
template <> struct TnLoader<bf16_t> {
    static __device__ __forceinline__ void load(Stage& s, const bf16_t* __restrict__ S, int64_t ld, int64_t ncols,
                                                int64_t Tred, int64_t c0, int64_t t0, int tid) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int item = tid + NT * i;
            const int cc = item & 15, tp = item >> 4;
            int64_t col = c0 + cc * 8;
            col = col <= ncols - 8 ? col : ncols - 8;
            const int64_t t = t0 + 2 * tp;
            u32x4 z = {0u, 0u, 0u, 0u};
            s.v[2 * i] = (t < Tred) ? *reinterpret_cast<const u32x4*>(S + t * ld + col) : z;
            s.v[2 * i + 1] = (t + 1 < Tred) ? *reinterpret_cast<const u32x4*>(S + (t + 1) * ld + col) : z;
        }
    }
    static __device__ __forceinline__ void store(const Stage& s, char* lds, int tid) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int item = tid + NT * i;
            const int cc = item & 15, tp = item >> 4;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t w0 = s.v[2 * i][e >> 1], w1 = s.v[2 * i + 1][e >> 1];
                const uint32_t lo = (e & 1) ? (w0 >> 16) : (w0 & 0xffffu);
                const uint32_t hi = (e & 1) ? (w1 & 0xffff0000u) : (w1 << 16);
                *reinterpret_cast<uint32_t*>(lds + lds_off(cc * 8 + e, tp * 4)) = lo | hi;
            }
        }
    }
};
template <> struct TnLoader<float> {
    static __device__ __forceinline__ void load(Stage& s, const float* __restrict__ S, int64_t ld, int64_t ncols,
                                                int64_t Tred, int64_t c0, int64_t t0, int tid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = tid + NT * i;
            const int cc = item & 31, t = item >> 5;
            int64_t col = c0 + cc * 4;
            col = col <= ncols - 4 ? col : ncols - 4;
            u32x4 z = {0u, 0u, 0u, 0u};
            s.v[i] = (t0 + t < Tred) ? *reinterpret_cast<const u32x4*>(S + (t0 + t) * ld + col) : z;
        }
    }
    static __device__ __forceinline__ void store(const Stage& s, char* lds, int tid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = tid + NT * i;
            const int cc = item & 31, t = item >> 5;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<uint32_t*>(lds + lds_off(cc * 4 + e, t * 4)) = s.v[i][e];
        }
    }
};

template <typename T>
__device__ __forceinline__ void compute_tile(const char* ldsA, const char* ldsB, f32x16 (&acc)[2][2],
                                             int wm, int wn, int lane) {
    typedef typename Chunk<T>::type chunk_t;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        chunk_t a[2], b[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            a[mi] = *reinterpret_cast<const chunk_t*>(ldsA + lds_off(wm * 64 + mi * 32 + l31, (2 * kk + h) * 16));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            b[ni] = *reinterpret_cast<const chunk_t*>(ldsB + lds_off(wn * 64 + ni * 32 + l31, (2 * kk + h) * 16));
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = mma_chunk(b[ni], a[mi], acc[mi][ni]);   // D[i = n][j = m]
    }
}

template <typename T, bool TN>
__global__ __launch_bounds__(NT) void gemm_g128_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: block b runs on XCD b % 8; give every XCD a contiguous range of tiles so
    // that the blocks sharing an activation row-panel hit the same L2.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = wgid / p.tiles_n, tn = wgid % p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    constexpr int KSTEP = ROWB / sizeof(T);
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    const int nk = (int)((p.K + KSTEP - 1) / KSTEP);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    Stage sa, sb;
    auto gload = [&](int t) {
        const int64_t k0 = (int64_t)t * KSTEP;
        if (TN) {
            TnLoader<T>::load(sa, A, p.lda, p.M, p.K, m0, k0, tid);
            TnLoader<T>::load(sb, B, p.ldb, p.N, p.K, n0, k0, tid);
        } else {
            load_nt<T>(sa, A, p.lda, p.M, p.K, m0, k0, tid);
            load_nt<T>(sb, B, p.ldb, p.N, p.K, n0, k0, tid);
        }
    };
    auto lstore = [&](int buf) {
        char* la = smem + buf * 2 * TILE_BYTES;
        char* lb = la + TILE_BYTES;
        if (TN) {
            TnLoader<T>::store(sa, la, tid);
            TnLoader<T>::store(sb, lb, tid);
        } else {
            store_nt(sa, la, tid);
            store_nt(sb, lb, tid);
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) gload(t + 1);
        compute_tile<T>(smem + buf * 2 * TILE_BYTES, smem + buf * 2 * TILE_BYTES + TILE_BYTES, acc, wm, wn, lane);
        if (t + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * h;
                if (n >= p.N) continue;
                f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                epilogue_quad(p, m, n, v);
            }
        }
    }
}

template <typename T, bool TN>
int launch_g128(const GemmParams& p, hipStream_t stream) {
    const size_t lds = 4 * TILE_BYTES;   // 64 KiB
    static OncePerDevice once;
    if (once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g128_kernel<T, TN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int64_t nblk = (int64_t)p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm_g128_kernel<T, TN>), dim3((unsigned)nblk), dim3(NT), lds, stream, p);
    ME_CHECK_LAUNCH("me_gemm(g128)");
    return ME_OK;
}

// fold the partial column sums of A (fixed order: deterministic).  The LAST workgroups of the grid take 32 columns each,
// eight threads per column striding over the partial rows, then a fixed-order fold through LDS -- kept off the first
// workgroups so it overlaps the slab fold instead of delaying it.
__device__ __forceinline__ void fold_colsum(const GemmParams& p, const float* __restrict__ cs_part, int n_part, float* __restrict__ colsum_out) {
    __shared__ float red[8][32];
    const int nblk = (int)((p.M + 31) / 32);
    const int nwg = (int)(gridDim.x * gridDim.y);
    const int b = nwg - 1 - (int)(blockIdx.y * gridDim.x + blockIdx.x);
    if (b < nblk || nwg < nblk) {
        for (int cb = b; cb < nblk; cb += nwg) {
            const int64_t m = (int64_t)cb * 32 + (threadIdx.x & 31);
            const int jg = threadIdx.x >> 5;
            float s = 0.f;
            // (balanced partition: the tn = 0 tile of this column's tile row left one partial row per part it was cut into)
            if (m < p.M) {
                if (p.sk_wgs) {
                    // levels: every (level, N-tile) row; leftover: row S tiles_n + j tiles_n + tn exists for the j-th leftover part of tile (tm, tn)
                    const int lv = p.sk_levels * p.tiles_n, t0 = (int)(m >> 8) * p.tiles_n;
                    for (int j = jg; j < n_part; j += 8) {
                        const bool ok = j < lv || (j - lv) / p.tiles_n < ME_SK_PARTS(p, t0 + (j - lv) % p.tiles_n) - p.sk_levels;
                        if (ok) s += cs_part[(int64_t)j * p.M + m];
                    }
                } else {
                    for (int j = jg; j < n_part; j += 8) s += cs_part[(int64_t)j * p.M + m];
                }
            }
            red[jg][threadIdx.x & 31] = s;
            __syncthreads();
            if (threadIdx.x < 32 && m < p.M) {
                float t = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x];
                colsum_out[m] = p.beta != 0.0f ? t + p.beta * colsum_out[m] : t;      // follows C's beta
            }
            __syncthreads();
        }
    }
}

// ---- split-K fold: sum the fp32 slabs [S][M][N] and apply the real epilogue.
// A WAVE owns 256 consecutive columns of one row (64 lanes x one quad), blockIdx.x walks the column blocks, blockIdx.y / the
// wave index the rows -- no division.  Slab reads go out in batches (8 / 4 / the last 1 .. 3 together, the surplus of the last
// batch masked), added in slab order: deterministic.
__device__ __forceinline__ f32x4 fold_slabs(const float* __restrict__ s0, int S, int64_t sstride) {
    f32x4 v = ME_NT_LOAD(ME_POL_SLAB, reinterpret_cast<const f32x4*>(s0));
    int s = 1;
    for (; s + 7 < S; s += 8) {
        f32x4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = ME_NT_LOAD(ME_POL_SLAB, reinterpret_cast<const f32x4*>(s0 + (int64_t)(s + j) * sstride));
#pragma unroll
        for (int j = 0; j < 8; ++j) v += t[j];
    }
    if (s + 3 < S) {
        f32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = ME_NT_LOAD(ME_POL_SLAB, reinterpret_cast<const f32x4*>(s0 + (int64_t)(s + j) * sstride));
#pragma unroll
        for (int j = 0; j < 4; ++j) v += t[j];
        s += 4;
    }
    if (s < S) {
        f32x4 t[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int sj = s + j < S ? s + j : S - 1;
            t[j] = ME_NT_LOAD(ME_POL_SLAB, reinterpret_cast<const f32x4*>(s0 + (int64_t)sj * sstride));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v += s + j < S ? t[j] : z;
        }
    }
    return v;
}

// The folds the encoder actually runs, one instantiation each -- NO run-time conditions around the operand loads: every `if (p.bias)
// v += load` of the generic epilogue costs a vmcnt(0) at its join (hipcc), i.e. one more serial memory round trip per operand, and
// hipcc turns a "load from the selected pointer" back into those branches.  PMC, B = 1 forward: 5 080 wait cycles per wave in the
// generic fold against 1 320 in round 1's (which had fewer operands to test for) -- 0.45 ms of a 0.9 ms forward.
//   BIAS  bias[n] present            RES  0 none / 1 bf16 / 2 fp32 residual row operand (plain rows)
//   CST   0 bf16 store / 1 fp32 store / 2 fp32 store + beta * C_old (accumulating weight gradient)
//   ACT   GELU (bf16-mode form by the output dtype, as the GEMM epilogues)       CS  fold the partial column sums too (bias gradient)
template <bool BIAS, int RES, int CST, bool ACT, bool CS>
__global__ __launch_bounds__(256) void splitk_fold_kernel(const GemmParams p, const float* __restrict__ slabs, int S, const float* __restrict__ cs_part,
                                                          int n_part, float* __restrict__ colsum_out, int64_t sstride) {
    if (CS) fold_colsum(p, cs_part, n_part, colsum_out);
    const int q = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63);
    if (q >= (int)(p.N / 4)) return;
    const int64_t n = (int64_t)q * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (BIAS) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    for (int64_t m = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6); m < p.M; m += (int64_t)gridDim.y * 4) {
        f32x4 r4 = {0.f, 0.f, 0.f, 0.f}, c4 = r4;
        u32x2 r2 = {0u, 0u};
        if (RES == 1) r2 = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(p.residual) + m * p.ldres + n);
        if (RES == 2) r4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.residual) + m * p.ldres + n);
        if (CST == 2) c4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.C) + m * p.ldc + n);
        // (balanced partition: this tile's own slab count; a wave's 256 columns are one tile's)
        const int St = p.sk_wgs ? ME_SK_PARTS(p, (int)(m >> 8) * p.tiles_n + (int)blockIdx.x) : S;
        f32x4 v = fold_slabs(slabs + m * p.N + n, St, sstride);
        v = v * p.alpha + bias4;
        if (ACT) v = gelu_for4(v, CST == 0 ? ME_BF16 : ME_F32);
        if (RES == 1) v += f32x4{__uint_as_float(r2[0] << 16), __uint_as_float(r2[0] & 0xffff0000u), __uint_as_float(r2[1] << 16), __uint_as_float(r2[1] & 0xffff0000u)};
        if (RES == 2) v += r4;
        if (CST == 2) v += p.beta * c4;
        if (CST == 0) {
            bf16x4 o;
            o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
            *reinterpret_cast<bf16x4*>(reinterpret_cast<uint16_t*>(p.C) + m * p.ldc + n) = o;
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + m * p.ldc + n) = v;
        }
    }
}

// ... and everything else the header allows (column scale, row remaps, saved pre-activation, gelu' operand, folded LayerNorm)
template <bool LIN>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, const float* __restrict__ slabs, int S,
                                                            const float* __restrict__ cs_part, int n_part,
                                                            float* __restrict__ colsum_out, int64_t sstride) {
    if (colsum_out) fold_colsum(p, cs_part, n_part, colsum_out);
    const int q = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63);
    if (q >= (int)(p.N / 4)) return;
    const int64_t n = (int64_t)q * 4;
    for (int64_t m = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6); m < p.M; m += (int64_t)gridDim.y * 4) {
        const int St = p.sk_wgs ? ME_SK_PARTS(p, (int)(m >> 8) * p.tiles_n + (int)blockIdx.x) : S;
        const f32x4 v = fold_slabs(slabs + m * p.N + n, St, sstride);
        if (LIN) epilogue_quad_lin(p, m, n, v);
        else epilogue_quad(p, m, n, v);
    }
}

static void launch_splitk_reduce(const GemmParams& p, unsigned /*nb*/, hipStream_t stream, const float* slabs, int S, const float* cs_part,
                                 int n_part, float* colsum_out, int64_t sstride = 0) {
    if (sstride == 0) sstride = p.M * p.N;
    const unsigned gx = (unsigned)((p.N / 4 + 63) / 64);
    int64_t gy = (p.M + 3) / 4;
    const int64_t cap = 4096 / gx > 1 ? 4096 / gx : 1;          // ~16 blocks per CU at most; rows beyond that by stride
    if (gy > cap) gy = cap;
    const dim3 grid(gx, (unsigned)gy);
#define ME_FOLD(BIAS, RES, CST, ACT, CS) \
    do { hipLaunchKernelGGL((splitk_fold_kernel<BIAS, RES, CST, ACT, CS>), grid, dim3(256), 0, stream, p, slabs, S, cs_part, n_part, colsum_out, sstride); return; } while (0)
    const bool simple = !p.preact && !p.aux && !p.row_affine && !p.colscale && !p.flags && p.res_row_mod == 0 && p.out_group_rows == 0 &&
                        !me_is_planes(p.c_dtype);
    const int res = !p.residual ? 0 : (p.res_dtype == ME_BF16 ? 1 : 2);
    if (simple) {
        const bool bias = p.bias != nullptr, gelu = p.act == ME_ACT_GELU;
        if (p.c_dtype == ME_F32 && !gelu && res == 0 && !bias) {          // weight gradients
            if (p.beta == 1.0f || p.beta == 0.0f) {
                if (p.beta != 0.0f) { if (colsum_out) ME_FOLD(false, 0, 2, false, true); else ME_FOLD(false, 0, 2, false, false); }
                else { if (colsum_out) ME_FOLD(false, 0, 1, false, true); else ME_FOLD(false, 0, 1, false, false); }
            }
        }
        if (p.beta == 0.0f && !colsum_out && bias) {                      // forward / dgrad launches split over the reduction (small batches, tails)
            if (p.c_dtype == ME_BF16) {
                if (gelu && res == 0) ME_FOLD(true, 0, 0, true, false);
                if (!gelu && res == 0) ME_FOLD(true, 0, 0, false, false);
                if (!gelu && res == 1) ME_FOLD(true, 1, 0, false, false);
            } else {
                if (gelu && res == 0) ME_FOLD(true, 0, 1, true, false);
                if (!gelu && res == 0) ME_FOLD(true, 0, 1, false, false);
                if (!gelu && res == 2) ME_FOLD(true, 2, 1, false, false);
            }
        }
        if (p.beta == 0.0f && !colsum_out && !bias && !gelu && res == 0) {   // dgrad (no bias)
            if (p.c_dtype == ME_BF16) ME_FOLD(false, 0, 0, false, false);
            else ME_FOLD(false, 0, 1, false, false);
        }
    }
#undef ME_FOLD
    if (p.act == ME_ACT_NONE && !p.preact && !p.aux && !p.row_affine)
        hipLaunchKernelGGL(splitk_reduce_kernel<true>, grid, dim3(256), 0, stream, p, slabs, S, cs_part, n_part, colsum_out, sstride);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<false>, grid, dim3(256), 0, stream, p, slabs, S, cs_part, n_part, colsum_out, sstride);
}

struct GemmPlan {
    int family;      // 0 = g128, 2 = g2b (two 4-wave workgroups / CU), 3 = g2w (8 waves, K-step 32), 4 = g3 (8 waves, K-tile 64)
    int bn;          // 256 / 128
    int bm;          // 256 / 128
    int kstep;       // reduction elements per pipeline step (32; g3: 64)
    int split_k;     // >= 1
    int ksteps_per_split;
    size_t ws_bytes;
    // NT tail split: the last tail_rows rows run as their own split-K problem (see plan_gemm)
    int64_t tail_rows;
    int tail_split, tail_ksteps;
    // g3 wgrad next to a communication kernel (me_gemm_reserve_cus): the balanced static partition over sk_wgs workgroups of sk_upt
    // K-tile pairs per tile (gemm3.hip, gemm_g3tn_sk_kernel); split_k then = the most slabs a tile gets
    int sk_wgs = 0, sk_upt = 0, sk_levels = 0, sk_l1 = 0;
};

// CUs a communication library's kernels hold while gradient buckets are reduced (me_gemm_reserve_cus; me_comm_init / _destroy set it)
std::atomic<int> g_reserved_cus{0};

#ifndef ME_SMALL_SPLIT_DEN
#define ME_SMALL_SPLIT_DEN 2                     // whole-problem split-K when tiles <= slots / this.  Measured on the reference's shapes
                                                 // (profiles/r04_small_split_ab.txt): never = +7..18 % per forward at B = 32..256 x N = 16..197,
                                                 // B = 1 0.92 -> 1.43 ms; / 3 -> / 2: M = 8 224 fwd + dX 6.9 -> 6.74 ms, the rest unchanged
#endif
#ifndef ME_SMALL_SPLIT_LONGK
#define ME_SMALL_SPLIT_LONGK 192
#endif
GemmPlan plan_gemm(const me_gemm_desc* d, const GemmParams& p, bool allow_sk = true) {
    GemmPlan pl{0, 0, 128, 0, 1, 0, 0, 0, 1, 0};
    const GemmDev dev = gemm_dev();
    const int ffam = dev.family, fbn = dev.bn;
    if (d->ab_dtype != ME_BF16 || ffam == 0) return pl;
    // default (measured on the encoder's shapes, tools/gemm_dev): NT with at least half a chip of 256x256 tiles -> g3
    // (K-tile 64, 4-phase ping-pong: 860-1000 TF on the Base shapes against 500-780 for the K-step-32 kernels); fewer
    // tiles or K not a multiple of 128 -> the K-step-32 kernels with their split-K forms; TN (wgrad) -> g2b.
    int fam = ffam > 0 ? ffam : 2;
    if (ffam < 0 && d->op == ME_GEMM_TN && d->M >= 256 && d->N >= 256 && d->K >= 4096) fam = 4;      // wgrad: g3 (950 vs 640 TF)
    if (ffam < 0 && d->op == ME_GEMM_NT && d->M >= 256 && d->N >= 256) {
        const int64_t t256 = ((d->M + 255) / 256) * ((d->N + 255) / 256);
        fam = (t256 >= 128 && g3_supported(p, d->op)) ? 4 : ((d->N <= 768 && d->K <= 1024) ? 2 : 3);
    }
    if (fam == 4 && d->op == ME_GEMM_NT && !g3_supported(p, d->op)) fam = 3;
    if (fam == 4 && d->op == ME_GEMM_TN && (!g3_tn_supported(p) || d->M < 256 || d->N < 256)) fam = 2;
    // (the g3 wgrad kernel takes any reduction length: rows past K read zeros through its descriptors -- a ragged token count,
    //  B x N not a multiple of 32, no longer sends the weight gradients of a whole Block to the generic kernel)
    if (!(fam == 4 && d->op == ME_GEMM_TN) && !g2b_supported(p, d->op)) return pl;
    if (ffam < 0 && (d->M < 128 || d->N < 128)) return pl;       // tiny problems: g128 is enough
    pl.family = fam;
    pl.bm = fam == 2 ? 128 : 256;
    if (kMeDev && fam == 2 && fbn == 64 && d->op == ME_GEMM_NT) pl.bm = 64;      // dev A/B: the 64-row small-M form
    // Small / mid-size NT problems (fewer than half a chip of 256 x 256 tiles: the reference's own batches, B = 32 x 96 .. 257
    // tokens, and B = 1 .. 32 inference): pick the TILE so that the launch has enough workgroups for the chip WITHOUT a K split --
    // the largest of 128 x 256 / 128 x 128 / 64 x 128 that still gives >= 140 tiles, else the smallest.  Measured on M = 197 ..
    // 6 304 (profiles/r05_small_gemm_plans.txt): 1.2 .. 4x faster per launch than the round-4 plans (256-wide tiles + whole-problem
    // split-K + fold), e.g. M = 3 072: qkv 36 -> 19 us, proj 23 -> 11 us, fc2 41 -> 31 us; M = 197: fc1 41 -> 10 us.
    bool small_nt = false;
    if (ffam < 0 && fam != 4 && d->op == ME_GEMM_NT && d->N % 128 == 0) {
        fam = 2; pl.family = 2; small_nt = true;
        const int64_t tm128 = (d->M + 127) / 128, tm64 = (d->M + 63) / 64;
        const int64_t c256 = tm128 * ((d->N + 255) / 256), c128 = tm128 * (d->N / 128), c64 = tm64 * (d->N / 128);
        (void)c64;
        if (c256 >= 140 && d->N % 256 == 0) { pl.bm = 128; pl.bn = 256; }
        else if (c128 >= 140) { pl.bm = 128; pl.bn = 128; }
        else { pl.bm = 64; pl.bn = 128; }
    }
    pl.kstep = 32;
    const int64_t tm = (d->M + pl.bm - 1) / pl.bm;
    const int64_t t256 = tm * ((d->N + 255) / 256), t128 = tm * ((d->N + 127) / 128);
    const int nk = (int)(d->K / pl.kstep);
    const int SLOTS = fam == 2 ? 512 : 256;                      // co-resident workgroups on the chip
    if (d->op == ME_GEMM_TN && fam == 4) {
        // wgrad on the g3 skeleton (gemm3.hip): 256 x 256 output tiles, the reduction split over the CUs in K-tile pairs
        if (!g3_tn_supported(p)) return GemmPlan{0, 0, 128, 0, 1, 0, 0, 0, 1, 0};
        pl.bn = 256; pl.bm = 256; pl.kstep = 64;
        const int64_t tiles = ((d->M + 255) / 256) * ((d->N + 255) / 256);
        const int nkt = (int)((d->K + 63) / 64);
        // workgroup slots: every CU, less the ones a communication kernel holds while this launch runs (me_gemm_reserve_cus) -- with
        // exactly one item per CU, ANY held CU sends the launch's last workgroups into a second round (profiles/r05_contention.txt)
        const int reserved = g_reserved_cus.load();
        const int slots = 256 - (reserved > 0 ? (reserved < 128 ? reserved : 128) : 0);
        int s = (int)(slots / tiles);
        if (s < 1) s = 1;
        int ktp = (nkt + s - 1) / s;
        ktp += ktp & 1;
        if (ktp < 2) ktp = 2;
        pl.ksteps_per_split = ktp;
        pl.split_k = (nkt + ktp - 1) / ktp;
        int slabs = pl.split_k;
        const int upt = (nkt + 1) / 2;                                    // K-tile pairs per tile
        const int S = (int)(slots / tiles), E = (int)(slots - S * tiles);
        if (reserved > 0 && S >= 1 && E > 0 && tiles * upt >= 2 * (int64_t)slots && tiles * upt < (1ll << 30)) {
            // slots that are not a multiple of the tile count: the uniform split would leave E = slots - S tiles of them idle (36 tiles on 240
            // slots: 6 parts of 132 K-tiles instead of 7 of 114 = +16 %).  Instead: S whole split levels of L1 pairs + the leftover of every
            // tile shared by the E extra workgroups (gemm3.hip, gemm_g3tn_sk_kernel): every workgroup carries ~tiles x pairs / slots
            // pairs per level part: the integer next to tiles x pairs / slots that leaves the extra workgroups no more than a level part carries
            // (rounding DOWN pushes S x the remainder into the leftover: 9 tiles x 394 pairs on 240 slots -> 14 per part but 45 per extra
            // workgroup, a 3x tail -- the first version's +24 % on a free GPU, profiles/r06_contention.txt)
            int L1 = (int)(tiles * upt / slots);
            {
                auto tail = [&](int l1) { return S * l1 > upt ? (int64_t)1 << 40 : std::max<int64_t>(l1, (tiles * (int64_t)(upt - S * l1) + E - 1) / E); };
                if (tail(L1 + 1) < tail(L1)) L1 += 1;
            }
            const int Ul = upt - S * L1;
            int smax = S;
            for (int t = 0; t < (int)tiles; ++t) smax = std::max(smax, S + sk_left_parts(t, E, Ul, tiles * (int64_t)Ul));
            if (allow_sk && L1 >= 1) { pl.sk_wgs = slots; pl.sk_upt = upt; pl.sk_levels = S; pl.sk_l1 = L1; pl.split_k = smax; }
            slabs = std::max(slabs, smax);                                // (the workspace query covers both forms)
        }
        pl.ws_bytes = (size_t)slabs * (size_t)g3_tn_slab_stride(d->M, d->N) * sizeof(float);
        if (d->colsum_a)                  // partial column sums of A: one [M] row per (split, N-tile) / per part of the tn = 0 tiles
            pl.ws_bytes += (size_t)slabs * (size_t)((d->N + 255) / 256) * (size_t)d->M * sizeof(float);
#if G3_TN_FOLD
        pl.ws_bytes += 256 + (size_t)((d->M + 255) / 256) * sizeof(unsigned);      // tile-row counters of the in-kernel fold (behind the rest, 256-byte aligned)
#endif
    } else if (d->op == ME_GEMM_TN) {
        pl.bn = fbn ? fbn : ((d->N % 256 == 0 || d->N > 512) ? 256 : 128);
        const int64_t tiles = pl.bn == 256 ? t256 : t128;
        int s = (int)(SLOTS / tiles);
        if (s < 1) s = 1;
        const int min_steps = 512 / pl.kstep;                   // keep >= 512 reduction rows per slice
        while (s > 1 && nk / s < min_steps) --s;
        pl.ksteps_per_split = (nk + s - 1) / s;
        pl.split_k = (nk + pl.ksteps_per_split - 1) / pl.ksteps_per_split;
        if (pl.split_k > 1) {
            pl.ws_bytes = (size_t)pl.split_k * (size_t)d->M * (size_t)d->N * sizeof(float);
            if (d->colsum_a && fam == 2)      // partial column sums of A: one [M] row per (split, N-tile)
                pl.ws_bytes += (size_t)pl.split_k * (size_t)((d->N + pl.bn - 1) / pl.bn) * (size_t)d->M * sizeof(float);
        }
    } else {
        if (fam >= 3) pl.bn = 256;
        else if (fbn) pl.bn = fbn == 64 ? 128 : fbn;
        else if (small_nt) {}                                    // (chosen above)
        else pl.bn = d->N > 128 ? 256 : 128;                     // measured: g2b_256 beats g2b_128 on every encoder shape
        pl.ksteps_per_split = nk;
        if (fam == 4) {
            // Tile quantisation (see gemm3.hip): when the last round is mostly empty its tiles run as `tail_split` parts each
            // inside the same launch (fp32 slabs + the deterministic fold, which also applies the epilogue).  Gated to narrow
            // outputs with a long reduction (N <= 1024, K >= 2048: fc2, the qkv / fc1 dgrads -- measured +2..6 %); at
            // K = 768 the fold's extra pass costs more than the idle third round (-15 %), and at N = 3072 one sparse round
            // in ten is cheap.
            const int64_t tn_ = (d->N + 255) / 256, tiles4 = tm * tn_;
            const int SL = 256;
            const int64_t R4 = tiles4 / SL, rem4 = tiles4 - R4 * SL;
            const int nkt = (int)(d->K / 64);
            // (not for the shapes the resident kernel takes: with its cheap tile seams the split's slabs + fold cost more
            // than the idle CUs of the last round -- measured 230 vs 241 us on fc2 forward, 170 vs 185 us on qkv dgrad)
            const bool resident = d->c_dtype == ME_BF16 && d->alpha == 1.0f && dev.g3_persistent == 1;
            if (!resident && d->N <= 1024 && d->K >= 2048 && dev.tail_split && R4 >= 1 && rem4 * 20 >= SL && rem4 * 10 <= SL * 6 && d->res_row_mod == 0 &&
                d->out_group_rows == 0) {
                const int64_t m_main = (R4 * SL) / tn_;
                const int64_t tail_tiles = (tm - m_main) * tn_;
                int s = (int)(SL / tail_tiles);
                int ktp = (nkt + s - 1) / s;
                ktp += ktp & 1;                                   // whole K-tile pairs
                if (s >= 2 && m_main >= 1 && ktp >= 2) {
                    pl.tail_rows = d->M - m_main * 256;
                    pl.tail_ksteps = ktp;
                    pl.tail_split = (nkt + ktp - 1) / ktp;
                    if (pl.tail_split >= 2) pl.ws_bytes = (size_t)pl.tail_split * (size_t)pl.tail_rows * (size_t)d->N * sizeof(float);
                    else pl.tail_rows = 0;
                }
            }
#ifdef ME_DEV
            // dev build: the persistent stream-K form (its scratch = one fp32 tile per CU)
            const int64_t tiles = tm * ((d->N + 255) / 256);
            if (tiles >= 128 && dev.g3_persistent == 2) { pl.ws_bytes = g3_workspace_bytes(); pl.tail_rows = 0; }
#endif
            return pl;
        }
        // Tile quantisation: T tiles on SLOTS co-resident workgroups take ceil(T / SLOTS) rounds; the encoder's N = 768
        // outputs give 591 tiles = 2.31 rounds -> 3 (23 % idle), N = 3072 gives 9.23 -> 10.  When the last round is
        // mostly empty, the rows of that round are carved off as a second problem whose reduction is split over the idle
        // workgroups (fp32 slabs + the deterministic fold that also applies the epilogue): 2 rounds + 1/3 instead of 3.
        // Measured: +4..5 % on the 256x256 kernel at N = 768 (fc2 forward, fc1 / qkv dgrad); a loss at N = 3072 (one
        // sparse round in ten is cheap: its workgroups run faster on an emptier chip) and on the two-workgroup kernel at
        // K = 768 (slices too short) -- hence the fam == 3 / N <= 1024 gate.
        const int64_t tn_ = (d->N + pl.bn - 1) / pl.bn, tiles = tm * tn_;
        const int64_t R = tiles / SLOTS, rem = tiles - R * SLOTS;
        if (fam == 3 && d->N <= 1024 && dev.tail_split && R >= 1 && rem * 20 >= SLOTS && rem * 10 <= SLOTS * 6 &&
            d->res_row_mod == 0 && d->out_group_rows == 0 && d->M % pl.bm == 0) {
            const int64_t m_main = (R * SLOTS) / tn_;
            const int64_t tail_tiles = (tm - m_main) * tn_;
            int s = (int)(SLOTS / tail_tiles);
            while (s > 1 && nk / s < 8) --s;                     // >= 8 K-steps (256 reduction elements) per slice
            if (s >= 2 && m_main >= 1) {
                pl.tail_rows = (tm - m_main) * pl.bm;
                pl.tail_ksteps = (nk + s - 1) / s;
                pl.tail_split = (nk + pl.tail_ksteps - 1) / pl.tail_ksteps;
                pl.ws_bytes = (size_t)pl.tail_split * (size_t)pl.tail_rows * (size_t)d->N * sizeof(float);
            }
        }
        // Small problems (small-batch inference: M = B * N rows with B = 1..32): fewer tiles than a third of the chip's
        // workgroup slots means the launch is pure latency -- one workgroup walks the whole reduction while 2/3 of the CUs
        // idle.  The WHOLE problem then runs split over the reduction (same slabs + fold as the tail split, m_main = 0).
        // (with the tile chosen by count -- small_nt -- a K split pays only for long reductions on a handful of tiles: B = 1 fc2 / dgrads)
        // ... or a very long one on up to half the slots: K >= 6 144 only occurs as the three-plane (ME_BF16X3) form of fc2 / the fc1 dgrad,
        // 144 tiles x 288 K-steps at the reference's M = 3 072 (ME_SMALL_SPLIT_LONGK K-steps of 32; measured below)
        if (dev.tail_split && pl.tail_rows == 0 && tiles * ME_SMALL_SPLIT_DEN <= SLOTS && nk >= (small_nt ? 64 : 16) &&
            (!small_nt || tiles <= 96 || nk >= ME_SMALL_SPLIT_LONGK)) {
            int s = (int)(SLOTS / tiles);
            while (s > 1 && nk / s < 8) --s;
            if (s >= 2) {
                pl.tail_rows = d->M;
                pl.tail_ksteps = (nk + s - 1) / s;
                pl.tail_split = (nk + pl.tail_ksteps - 1) / pl.tail_ksteps;
                pl.ws_bytes = (size_t)pl.tail_split * (size_t)d->M * (size_t)d->N * sizeof(float);
            }
        }
    }
    return pl;
}

int fill_params(const me_gemm_desc* d, GemmParams& p) {
    ME_CHECK_ARG(d != nullptr, "me_gemm: null descriptor");
    ME_CHECK_ARG(d->op == ME_GEMM_NT || d->op == ME_GEMM_TN, "me_gemm: bad op %d", d->op);
    ME_CHECK_ARG(me_dtype_ok(d->ab_dtype) && (me_out_dtype_ok(d->c_dtype) || d->c_dtype == ME_BF16X2), "me_gemm: bad dtype");
    if (me_is_planes(d->c_dtype))     // fp32 result written as bf16 planes: the A operand of the next fp32-accurate Linear
        ME_CHECK_ARG(d->op == ME_GEMM_NT && d->beta == 0.0f && d->ldc >= (d->c_dtype == ME_BF16X3 ? 3 : 2) * d->N && d->out_group_rows == 0 && !d->colsum_a,
                     "me_gemm: ME_BF16X3 / ME_BF16X2 output: NT, beta = 0, ldc >= 3 N / 2 N, plain rows");
    ME_CHECK_ARG(d->a_wrap_k == 0 || (d->op == ME_GEMM_NT && d->ab_dtype == ME_BF16 && d->a_wrap_k > 0 && d->a_wrap_k % 128 == 0 && d->a_wrap_k < d->K &&
                                      d->K - d->a_wrap_k <= d->a_wrap_k && d->lda >= d->a_wrap_k),
                 "me_gemm: a_wrap_k: NT, bf16, a multiple of 128 with K / 2 <= a_wrap_k < K, lda >= a_wrap_k");
    ME_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "me_gemm: empty problem M=%lld N=%lld K=%lld",
                 (long long)d->M, (long long)d->N, (long long)d->K);
    ME_CHECK_ARG(d->A && d->B && d->C, "me_gemm: null operand");
    const int E = d->ab_dtype == ME_BF16 ? 8 : 4;
    ME_CHECK_ARG(d->N % 4 == 0, "me_gemm: N=%lld must be a multiple of 4", (long long)d->N);
    ME_CHECK_ARG(d->ldc % 4 == 0, "me_gemm: ldc must be a multiple of 4");
    if (d->op == ME_GEMM_NT) {
        ME_CHECK_ARG(d->K % E == 0 && d->lda % E == 0 && d->ldb % E == 0,
                     "me_gemm(NT): K, lda, ldb must be multiples of %d elements (16 bytes)", E);
    } else {
        ME_CHECK_ARG(d->M % E == 0 && d->N % E == 0 && d->lda % E == 0 && d->ldb % E == 0,
                     "me_gemm(TN): M, N, lda, ldb must be multiples of %d elements (16 bytes)", E);
    }
    ME_CHECK_ARG(((uintptr_t)d->A | (uintptr_t)d->B | (uintptr_t)d->C) % 16 == 0, "me_gemm: operands must be 16-byte aligned");
    ME_CHECK_ARG(d->act == ME_ACT_NONE || d->act == ME_ACT_GELU, "me_gemm: bad act");
    ME_CHECK_ARG((d->flags & ~(ME_GEMM_SAVE_GELU_GRAD | ME_GEMM_AUX_IS_FACTOR)) == 0, "me_gemm: unknown flags");
    ME_CHECK_ARG(!(d->flags & ME_GEMM_SAVE_GELU_GRAD) || (d->act == ME_ACT_GELU && d->preact), "me_gemm: ME_GEMM_SAVE_GELU_GRAD needs act = GELU and preact");
    ME_CHECK_ARG(!(d->flags & ME_GEMM_AUX_IS_FACTOR) || d->aux, "me_gemm: ME_GEMM_AUX_IS_FACTOR needs aux");
    // (ME_GG8: the saved gelu' in eight bits -- only as the flagged factor, 8-byte rows; see me_gemm_takes_gg8)
    if (d->preact) ME_CHECK_ARG((me_dtype_ok(d->preact_dtype) && d->ldpre % 4 == 0) ||
                                (d->preact_dtype == ME_GG8 && d->flags == ME_GEMM_SAVE_GELU_GRAD && d->ldpre % 8 == 0 && (uintptr_t)d->preact % 8 == 0), "me_gemm: bad preact");
    if (d->aux) ME_CHECK_ARG((me_dtype_ok(d->aux_dtype) && d->ldaux % 4 == 0) ||
                             (d->aux_dtype == ME_GG8 && d->flags == ME_GEMM_AUX_IS_FACTOR && d->ldaux % 8 == 0 && (uintptr_t)d->aux % 8 == 0), "me_gemm: bad aux");
    if (d->residual) ME_CHECK_ARG(me_dtype_ok(d->res_dtype) && d->ldres % 4 == 0, "me_gemm: bad residual");
    p.A = d->A; p.B = d->B; p.C = d->C;
    p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
    p.c_dtype = d->c_dtype; p.act = d->act; p.alpha = d->alpha; p.beta = d->beta;
    p.bias = d->bias; p.colscale = d->colscale;
    p.preact = d->preact; p.ldpre = d->ldpre; p.preact_dtype = d->preact_dtype;
    p.flags = d->flags;
    p.aux = d->aux; p.ldaux = d->ldaux; p.aux_dtype = d->aux_dtype;
    p.residual = d->residual; p.ldres = d->ldres; p.res_dtype = d->res_dtype;
    p.res_row_mod = d->res_row_mod; p.out_group_rows = d->out_group_rows;
    p.out_group_stride = d->out_group_stride; p.out_row_offset = d->out_row_offset;
    p.split_k = 1; p.ksteps_per_split = 0;
    p.colsum_ws = nullptr;
    ME_CHECK_ARG((d->row_affine == nullptr && d->row_parts == nullptr) == (d->col_shift == nullptr), "me_gemm: row_affine (or row_parts) and col_shift go together");
    ME_CHECK_ARG(!d->row_affine || d->op == ME_GEMM_NT, "me_gemm: row_affine (folded LayerNorm) is defined for ME_GEMM_NT");
    p.row_affine = d->row_affine; p.col_shift = d->col_shift;
    p.row_nparts = 0; p.row_eps = 0.0f;
    p.sk_wgs = 0; p.sk_upt = 0; p.sk_levels = 0; p.sk_l1 = 0;
    p.a_wrap_kt = (int)(d->a_wrap_k / 64);
    if (d->row_parts) {      // the same fold, its pairs formed in the kernel from a previous launch's row_stats partials
        ME_CHECK_ARG(!d->row_affine && d->col_shift && d->op == ME_GEMM_NT, "me_gemm: row_parts replaces row_affine (NT, with col_shift)");
        ME_CHECK_ARG(d->row_nparts >= 1 && d->row_nparts <= 4 && (int64_t)d->row_nparts * ME_STATS_GROUP == d->K && d->row_eps >= 0.0f,
                     "me_gemm: row_parts: row_nparts = K / 256, 1 .. 4 (see me_gemm_takes_row_parts)");
        ME_CHECK_ARG((uintptr_t)d->row_parts % 8 == 0, "me_gemm: row_parts must be 8-byte aligned");
        p.row_affine = d->row_parts; p.row_nparts = d->row_nparts; p.row_eps = d->row_eps;
    }
    ME_CHECK_ARG(!d->row_stats || (d->op == ME_GEMM_NT && d->residual && (uintptr_t)d->row_stats % 8 == 0),
                 "me_gemm: row_stats goes with ME_GEMM_NT and a residual operand (8-byte aligned)");
    p.row_stats = d->row_stats;
    p.tn_colsum_out = nullptr;
    p.g3_full_tiles = 0; p.g3_split = 0; p.g3_ktp = 0; p.g3_slabs = nullptr; p.slab_stride = 0; p.g3_tickets = nullptr; p.g3_half = 0; p.g3_colgroups = 1;
    if (d->colsum_a) ME_CHECK_ARG(d->op == ME_GEMM_TN, "me_gemm: colsum_a is defined for ME_GEMM_TN only");
    p.debug = gemm_dev().debug;
    p.tiles_m = (int)((d->M + BM - 1) / BM);
    p.tiles_n = (int)((d->N + BN - 1) / BN);
    ME_CHECK_ARG((int64_t)p.tiles_m * p.tiles_n < (1ll << 31), "me_gemm: too many tiles");
    return ME_OK;
}

}  // namespace

// (patch_embed.hip: the same validation and parameter block for the projection it runs on its own kernel)
int gemm_fill_params(const me_gemm_desc* d, GemmParams& p) { return fill_params(d, p); }

extern "C" size_t me_gemm_workspace_bytes(const me_gemm_desc* d) {
    GemmParams p;
    if (fill_params(d, p) != ME_OK) return 0;
    return plan_gemm(d, p).ws_bytes;
}

extern "C" int me_gemm_fuses_colsum(const me_gemm_desc* d) {
    GemmParams p;
    if (!d || d->op != ME_GEMM_TN || fill_params(d, p) != ME_OK) return 0;
    const GemmPlan pl = plan_gemm(d, p);
    return (pl.family == 2 && pl.split_k > 1) || pl.family == 4;
}

extern "C" int me_gemm_emits_row_stats(const me_gemm_desc* d) {
    GemmParams p;
    if (!d || d->op != ME_GEMM_NT || d->ab_dtype != ME_BF16 || !d->residual) return 0;
    me_gemm_desc e = *d;
    e.row_stats = nullptr;
    if (fill_params(&e, p) != ME_OK) return 0;
    const GemmPlan pl = plan_gemm(&e, p);
    if (pl.family != 4 || pl.tail_rows > 0) return 0;
    p.tiles_m = (int)((d->M + 255) / 256);
    p.tiles_n = (int)((d->N + 255) / 256);
    return g3_emits_row_stats(p) ? 1 : 0;
}

extern "C" int me_gemm_takes_a_wrap(const me_gemm_desc* d) {
    GemmParams p;
    if (!d || d->op != ME_GEMM_NT || d->ab_dtype != ME_BF16 || !d->a_wrap_k || fill_params(d, p) != ME_OK) return 0;
    const GemmPlan pl = plan_gemm(d, p);
    const int e = pick_epi_ex(p);
    return (pl.family == 4 && (e == 0 || e == 4 || e == 8)) ? 1 : 0;
}

extern "C" int me_gemm_reserve_cus(int cus) {
    const int c = cus < 0 ? 0 : (cus > 128 ? 128 : cus);
    return g_reserved_cus.exchange(c);
}

extern "C" int me_gemm_takes_row_parts(const me_gemm_desc* d) {
    GemmParams p;
    if (!d || d->op != ME_GEMM_NT || d->ab_dtype != ME_BF16 || !d->row_parts || fill_params(d, p) != ME_OK) return 0;
    const GemmPlan pl = plan_gemm(d, p);
    if (pl.family != 4 || pl.tail_rows > 0) return 0;
    return g3_takes_row_parts(p) ? 1 : 0;
}

extern "C" int me_gemm_takes_gg8(const me_gemm_desc* d) {
    GemmParams p;
    if (!d || d->op != ME_GEMM_NT || d->ab_dtype != ME_BF16 || fill_params(d, p) != ME_OK) return 0;
    if (!((d->preact && d->preact_dtype == ME_GG8) || (d->aux && d->aux_dtype == ME_GG8))) return 0;
    const GemmPlan pl = plan_gemm(d, p);
    if (pl.family != 4 || pl.tail_rows > 0) return 0;
    return g3_takes_gg8(p) ? 1 : 0;
}

namespace {
// tn_launch: a replacement for launch_g3_tn (patch_embed.hip: the wgrad kernel that gathers its B operand from the image); with it,
// a problem the planner does not give to the g3 wgrad family is refused (ME_ERR_UNSUPPORTED) instead of run
typedef int (*TnLaunch)(const GemmParams& p, hipStream_t stream, const void* ctx);
int gemm_impl(const me_gemm_desc* d, hipStream_t stream, int* plan_out, TnLaunch tn_launch = nullptr, const void* tn_ctx = nullptr);
}

extern "C" int me_gemm(const me_gemm_desc* d, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ProfScope prof(d ? d->op : 0, d ? d->ab_dtype : 0, d ? d->M : 0, d ? d->N : 0, d ? d->K : 0, stream);
    return gemm_impl(d, stream, &prof.plan);
}

namespace {
int gemm_impl(const me_gemm_desc* d, hipStream_t stream, int* plan_out, TnLaunch tn_launch, const void* tn_ctx) {
    GemmParams p;
    int rc = fill_params(d, p);
    if (rc) return rc;
    GemmPlan pl = plan_gemm(d, p, tn_launch == nullptr);      // (the custom wgrad launchers -- patch embed, three planes -- walk uniform splits)
    if (tn_launch && !(pl.family == 4 && d->op == ME_GEMM_TN && !G3_TN_FOLD)) return ME_ERR_UNSUPPORTED;
    {   // me_gemm_profile_rec.plan
        const bool have_ws = pl.ws_bytes && d->workspace && (size_t)d->workspace_bytes >= pl.ws_bytes;
        const int parts = pl.split_k > 1 ? pl.split_k : (pl.tail_rows > 0 ? pl.tail_split : 1);
        *plan_out = (pl.family & 15) | ((have_ws && parts > 1) ? 16 : 0) | (pl.sk_wgs ? 32 : 0) | ((have_ws ? parts : 1) << 8);
    }
    if (d->colsum_a)
        ME_CHECK_ARG(((pl.family == 2 && pl.split_k > 1) || pl.family == 4) && d->workspace && (size_t)d->workspace_bytes >= pl.ws_bytes,
                     "me_gemm: colsum_a needs the split-K wgrad kernel and its workspace (see me_gemm_fuses_colsum)");
    if (d->row_stats)
        ME_CHECK_ARG(pl.family == 4 && d->op == ME_GEMM_NT && pl.tail_rows == 0, "me_gemm: row_stats is not available for this problem (see me_gemm_emits_row_stats)");
    if (d->a_wrap_k)
        ME_CHECK_ARG(pl.family == 4 && d->op == ME_GEMM_NT && (pick_epi_ex(p) == 0 || pick_epi_ex(p) == 4 || pick_epi_ex(p) == 8),
                     "me_gemm: a_wrap_k is not available for this problem (see me_gemm_takes_a_wrap)");
    if (d->row_parts)
        ME_CHECK_ARG(pl.family == 4 && d->op == ME_GEMM_NT && pl.tail_rows == 0 && g3_takes_row_parts(p),
                     "me_gemm: row_parts is not available for this problem (see me_gemm_takes_row_parts)");
    if ((d->preact && d->preact_dtype == ME_GG8) || (d->aux && d->aux_dtype == ME_GG8))
        ME_CHECK_ARG(pl.family == 4 && d->op == ME_GEMM_NT && pl.tail_rows == 0 && g3_takes_gg8(p),
                     "me_gemm: ME_GG8 is not available for this problem (see me_gemm_takes_gg8)");
    if (pl.family >= 1) {
        if (pl.family == 4 && d->op == ME_GEMM_TN) {
            ME_CHECK_ARG(d->workspace && (size_t)d->workspace_bytes >= pl.ws_bytes,
                         "me_gemm(TN, g3): workspace of me_gemm_workspace_bytes() required");
            p.tiles_m = (int)((d->M + 255) / 256);
            p.tiles_n = (int)((d->N + 255) / 256);
            GemmParams ps = p;
            ps.C = d->workspace;
            ps.split_k = pl.split_k;
            ps.ksteps_per_split = pl.ksteps_per_split;
            ps.slab_stride = g3_tn_slab_stride(d->M, d->N);
            if (d->colsum_a) ps.colsum_ws = reinterpret_cast<float*>(d->workspace) + (size_t)pl.split_k * (size_t)ps.slab_stride;
#if G3_TN_FOLD
            if (g3_tn_fold_ok(p, pl.split_k)) {
                // the fold inside the launch: the kernel writes C itself (gemm3.hip, gemm_g3tn_kernel<true>)
                GemmParams pf = p;
                pf.split_k = pl.split_k;
                pf.ksteps_per_split = pl.ksteps_per_split;
                pf.slab_stride = ps.slab_stride;
                pf.g3_slabs = reinterpret_cast<float*>(d->workspace);
                pf.colsum_ws = ps.colsum_ws;
                pf.tn_colsum_out = d->colsum_a;
                const size_t used = (size_t)pl.split_k * (size_t)ps.slab_stride * sizeof(float) +
                                    (d->colsum_a ? (size_t)pl.split_k * (size_t)p.tiles_n * (size_t)d->M * sizeof(float) : 0);
                pf.g3_tickets = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(d->workspace) + ((used + 255) & ~(size_t)255));
                return launch_g3_tn_fold(pf, stream);
            }
#endif
            if (pl.sk_wgs) {
                ps.sk_wgs = pl.sk_wgs; ps.sk_upt = pl.sk_upt; ps.sk_levels = pl.sk_levels; ps.sk_l1 = pl.sk_l1;
                rc = launch_g3_tn_sk(ps, stream);
                p.sk_wgs = pl.sk_wgs; p.sk_upt = pl.sk_upt; p.sk_levels = pl.sk_levels; p.sk_l1 = pl.sk_l1;      // (the fold sums each tile's own number of slabs)
            } else {
                rc = tn_launch ? tn_launch(ps, stream, tn_ctx) : launch_g3_tn(ps, stream);
            }
            if (rc) return rc;
            p.split_k = 1;
            const int64_t quads = d->M * (d->N / 4);
            int64_t nb = (quads + 255) / 256;
            if (nb > 2048) nb = 2048;
            launch_splitk_reduce(p, (unsigned)nb, stream, reinterpret_cast<const float*>(d->workspace), pl.split_k, ps.colsum_ws, pl.split_k * p.tiles_n,
                               d->colsum_a, ps.slab_stride);
            ME_CHECK_LAUNCH("me_gemm(g3 tn fold)");
            return ME_OK;
        }
        if (pl.family == 4) {
            p.tiles_m = (int)((d->M + 255) / 256);
            p.tiles_n = (int)((d->N + 255) / 256);
            p.split_k = 1;
            const bool have_ws = pl.ws_bytes && d->workspace && (size_t)d->workspace_bytes >= pl.ws_bytes;
            p.g3_full_tiles = p.tiles_m * p.tiles_n; p.g3_split = 1; p.g3_ktp = 0; p.g3_slabs = nullptr;
            if (pl.tail_rows > 0 && have_ws) {
                const int64_t m1 = d->M - pl.tail_rows;          // rows covered by whole tiles (a multiple of 256)
                p.g3_full_tiles = (int)(m1 / 256) * p.tiles_n;
                p.g3_split = pl.tail_split;
                p.g3_ktp = pl.tail_ksteps;
                p.g3_slabs = reinterpret_cast<float*>(d->workspace);
                rc = launch_g3(p, pick_epi_ex(p), nullptr, stream);
                if (rc) return rc;
                GemmParams pt = p;                               // the fold sees the tail rows as its own problem
                pt.M = pl.tail_rows;
                pt.split_k = 1;
                pt.A = nullptr;
                pt.C = reinterpret_cast<char*>(p.C) + (size_t)m1 * p.ldc * me_dtype_size(p.c_dtype);
                if (p.preact) pt.preact = reinterpret_cast<char*>(p.preact) + (size_t)m1 * p.ldpre * me_dtype_size(p.preact_dtype);
                if (p.aux) pt.aux = reinterpret_cast<const char*>(p.aux) + (size_t)m1 * p.ldaux * me_dtype_size(p.aux_dtype);
                if (p.residual) pt.residual = reinterpret_cast<const char*>(p.residual) + (size_t)m1 * p.ldres * me_dtype_size(p.res_dtype);
                if (p.row_affine) pt.row_affine = p.row_affine + 2 * m1;
                const int64_t quads = pt.M * (d->N / 4);
                int64_t nb = (quads + 255) / 256;
                if (nb > 2048) nb = 2048;
                launch_splitk_reduce(pt, (unsigned)nb, stream, reinterpret_cast<const float*>(d->workspace), pl.tail_split, nullptr, 0, nullptr);
                ME_CHECK_LAUNCH("me_gemm(g3 tail fold)");
                return ME_OK;
            }
#ifdef ME_DEV
            void* ws = (pl.tail_rows == 0 && have_ws) ? d->workspace : nullptr;
#else
            void* ws = nullptr;
#endif
            return launch_g3(p, pick_epi_ex(p), ws, stream);
        }
        auto run = [&](const GemmParams& q) { return launch_g2b(q, d->op, pl.bm, pl.bn, stream); };
        p.tiles_m = (int)((d->M + pl.bm - 1) / pl.bm);
        p.tiles_n = (int)((d->N + pl.bn - 1) / pl.bn);
        p.ksteps_per_split = pl.ksteps_per_split;
        if (pl.split_k > 1 && d->workspace && (size_t)d->workspace_bytes >= pl.ws_bytes) {
            GemmParams ps = p;
            ps.C = d->workspace;
            ps.split_k = pl.split_k;
            const int n_part = pl.split_k * p.tiles_n;
            if (d->colsum_a)
                ps.colsum_ws = reinterpret_cast<float*>(d->workspace) + (size_t)pl.split_k * (size_t)d->M * (size_t)d->N;
            rc = run(ps);
            if (rc) return rc;
            const int64_t quads = d->M * (d->N / 4);
            int64_t nb = (quads + 255) / 256;
            if (nb > 2048) nb = 2048;
            launch_splitk_reduce(p, (unsigned)nb, stream, reinterpret_cast<const float*>(d->workspace), pl.split_k, ps.colsum_ws, n_part,
                               d->colsum_a);
            ME_CHECK_LAUNCH("me_gemm(splitk reduce)");
            return ME_OK;
        }
        p.split_k = 1;
        p.ksteps_per_split = (int)(d->K / pl.kstep);
        if (pl.tail_rows > 0 && d->workspace && (size_t)d->workspace_bytes >= pl.ws_bytes) {
            const int64_t m1 = d->M - pl.tail_rows;
            if (m1 > 0) {
                GemmParams pm = p;                   // main part: rows [0, m1), fused epilogue as usual
                pm.M = m1;
                pm.tiles_m = (int)(m1 / pl.bm);
                rc = run(pm);
                if (rc) return rc;
            }
            GemmParams pt = p;                       // tail: rows [m1, M) -- every row-indexed operand moves down by m1
            pt.M = pl.tail_rows;
            pt.tiles_m = (int)((pl.tail_rows + pl.bm - 1) / pl.bm);
            pt.A = reinterpret_cast<const char*>(p.A) + (size_t)m1 * p.lda * me_dtype_size(d->ab_dtype);
            pt.C = reinterpret_cast<char*>(p.C) + (size_t)m1 * p.ldc * me_dtype_size(p.c_dtype);
            if (p.preact) pt.preact = reinterpret_cast<char*>(p.preact) + (size_t)m1 * p.ldpre * me_dtype_size(p.preact_dtype);
            if (p.aux) pt.aux = reinterpret_cast<const char*>(p.aux) + (size_t)m1 * p.ldaux * me_dtype_size(p.aux_dtype);
            if (p.residual) pt.residual = reinterpret_cast<const char*>(p.residual) + (size_t)m1 * p.ldres * me_dtype_size(p.res_dtype);
            if (p.row_affine) pt.row_affine = p.row_affine + 2 * m1;
            GemmParams ps = pt;                      // the split launch writes slabs [S][tail_rows][N]
            ps.C = d->workspace;
            ps.split_k = pl.tail_split;
            ps.ksteps_per_split = pl.tail_ksteps;
            rc = run(ps);
            if (rc) return rc;
            const int64_t quads = pt.M * (d->N / 4);
            int64_t nb = (quads + 255) / 256;
            if (nb > 2048) nb = 2048;
            launch_splitk_reduce(pt, (unsigned)nb, stream, reinterpret_cast<const float*>(d->workspace), pl.tail_split, nullptr, 0, nullptr);
            ME_CHECK_LAUNCH("me_gemm(tail fold)");
            return ME_OK;
        }
        return run(p);
    }
    if (d->ab_dtype == ME_BF16)
        return d->op == ME_GEMM_NT ? launch_g128<bf16_t, false>(p, stream) : launch_g128<bf16_t, true>(p, stream);
    return d->op == ME_GEMM_NT ? launch_g128<float, false>(p, stream) : launch_g128<float, true>(p, stream);
}
}  // namespace

// (patch_embed.hip) me_gemm for a g3 wgrad problem with the split-K kernel launch replaced; 1 / 0: would the planner take that route
int gemm_tn_with_launcher(const me_gemm_desc* d, hipStream_t stream, int (*launch)(const GemmParams&, hipStream_t, const void*), const void* ctx) {
    ProfScope prof(d ? d->op : 0, d ? d->ab_dtype : 0, d ? d->M : 0, d ? d->N : 0, d ? d->K : 0, stream);
    return gemm_impl(d, stream, &prof.plan, launch, ctx);
}
int gemm_tn_is_g3(const me_gemm_desc* d) {
    GemmParams p;
    if (!d || d->op != ME_GEMM_TN || fill_params(d, p) != ME_OK) return 0;
    return (plan_gemm(d, p).family == 4 && !G3_TN_FOLD) ? 1 : 0;
}
