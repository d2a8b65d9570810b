// layernorm.hip -- LayerNorm forward / backward over the channel dim of [rows, C] tokens.
//
// Replaces nn.LayerNorm(C, eps) (Block.norm1 / norm2, PointCloud/openpoints/models/layers/attention.py:46,50
// -> norm.py:65).  HBM-bound: one wavefront per token row, 8/16-byte vector loads, the row is held in
// registers between the statistics pass and the normalise pass (one HBM read, one write), statistics by
// wavefront xor-shuffles in fp32 with the two-pass (mean, then centred sum of squares) formulation.
#include "common.h"
#include <atomic>
#include <type_traits>

namespace {

constexpr int LN_THREADS = 256;           // 4 rows per block
constexpr int LN_MAX_VEC = 16;            // supports C up to 4*64*16 = 4096 on the vector path

// compile-time typed 4-element accesses: a runtime dtype switch around every load makes hipcc place a full
// s_waitcnt vmcnt(0) at each branch join, which serialises the 18 loads a wave has in flight per iteration.
template <typename T> __device__ __forceinline__ f32x4 ld4(const void* base, int64_t idx);
template <> __device__ __forceinline__ f32x4 ld4<float>(const void* base, int64_t idx) {
    return ME_NT_LOAD(ME_POL_LN_LD, reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx));
}
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const void* base, int64_t idx) {
    const u32x2 raw = ME_NT_LOAD(ME_POL_LN_LD, reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(base) + idx));
    return f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u), __uint_as_float(raw[1] << 16),
                 __uint_as_float(raw[1] & 0xffff0000u)};
}
template <typename T> __device__ __forceinline__ void st4(void* base, int64_t idx, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(void* base, int64_t idx, f32x4 v) {
    ME_NT_STORE(ME_POL_LN_ST, v, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx));
}
template <> __device__ __forceinline__ void st4<bf16_t>(void* base, int64_t idx, f32x4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    ME_NT_STORE(ME_POL_LN_ST, __builtin_bit_cast(u32x2, o), reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(base) + idx));
}

struct me_split3_tag {};                  // TY of ln_fwd_kernel: the normalised row goes out as ME_BF16X3 (three bf16 planes, [hi | lo | hi])

// VPL = number of 4-element vectors per lane (C = 256 * VPL on the fast path); dtypes are template parameters for the
// reason given above (a runtime switch around the row loads would serialise them).
// A wave takes LN_R = 2 rows per pass with EVERY load of the pass issued up front -- both rows and (forward) the affine vectors:
// with one row per wave and gamma / beta fetched behind the reductions a row was two dependent memory latencies and 1.5 KB in
// flight (round 3: 0.58 of the HBM peak).
constexpr int ln_rows_per_wave(int vpl) { return vpl <= 4 ? 2 : 1; }       // (wider rows: the registers go to the row itself)
template <int VPL, typename TX, typename TY>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, void* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int64_t rows, int C, float eps) {
    constexpr int LN_R = ln_rows_per_wave(VPL);
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6)) * LN_R;
    if (row0 >= rows) return;
    f32x4 v[LN_R][VPL], g[VPL], b[VPL];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;          // (a clamped re-read instead of a branch around loads)
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[r][i] = ld4<TX>(x, row * C + (int64_t)(lane + 64 * i) * 4);
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        g[i] = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * 4);
        b[i] = *reinterpret_cast<const f32x4*>(beta + (lane + 64 * i) * 4);
    }
    float mean[LN_R], rstd[LN_R];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) s += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
        mean[r] = wave_sum(s) / (float)C;
    }
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[r][i][e] - mean[r];
                q += d * d;
            }
        }
        rstd[r] = rsqrtf(wave_sum(q) / (float)C + eps);
    }
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        const int64_t row = row0 + r;
        if (row >= rows) break;
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean[r];
            if (rstd_out) rstd_out[row] = rstd[r];
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (lane + 64 * i) * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[r][i][e] - mean[r]) * rstd[r] * g[i][e] + b[i][e];
            if constexpr (std::is_same<TY, me_split3_tag>::value) store4_split3(reinterpret_cast<uint16_t*>(y) + row * 3 * C, C, c, o);
            else st4<TY>(y, row * C + c, o);
        }
    }
}

// ---- row statistics only (inference with the LayerNorm FOLDED into the next Linear, see me_row_stats in include/metaenc.h):
// the same rows-in-registers two-pass statistics as ln_fwd_kernel, but nothing is normalised or written back -- per row the
// pair (rstd, -rstd * mean) that the folded GEMM's epilogue applies.  Reads rows * C elements once.
template <int VPL, typename TX>
__global__ __launch_bounds__(LN_THREADS) void ln_stats_kernel(const void* __restrict__ x, float* __restrict__ out, int64_t rows, int C, float eps) {
    constexpr int LN_R = ln_rows_per_wave(VPL);
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6)) * LN_R;
    if (row0 >= rows) return;
    f32x4 v[LN_R][VPL];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[r][i] = ld4<TX>(x, row * C + (int64_t)(lane + 64 * i) * 4);
    }
    float mean[LN_R];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) s += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
        mean[r] = wave_sum(s) / (float)C;
    }
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[r][i][e] - mean[r];
                q += d * d;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        if (lane == 0 && row0 + r < rows) *reinterpret_cast<float2*>(out + (row0 + r) * 2) = float2{rstd, -rstd * mean[r]};
    }
}
__global__ __launch_bounds__(LN_THREADS) void ln_stats_generic_kernel(const void* __restrict__ x, int x_dt, float* __restrict__ out,
                                                                      int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += load1_as_f32(x, x_dt, row * C + c);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = load1_as_f32(x, x_dt, row * C + c) - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) *reinterpret_cast<float2*>(out + row * 2) = float2{rstd, -rstd * mean};
}

// ---- the pairs of ln_stats_kernel from 256-column partials (mean_i, M2_i) a GEMM epilogue left behind ([nparts][rows] pairs,
// me_gemm_desc.row_stats; round 6: one pair per row and 256-column tile, it was one per 64-column wave column): one thread per row,
// the partials of consecutive rows are consecutive in memory.  Chan et al.'s combination for equal group sizes; exact in the sense
// that no large numbers are subtracted anywhere (the group M2 are sums of squared deviations already).  The folded qkv / fc1 GEMMs
// do the same arithmetic in their own epilogue where they can (me_gemm_desc.row_parts); this kernel serves the shapes they cannot.
__global__ __launch_bounds__(256) void row_stats_combine_kernel(const float2* __restrict__ part, int nparts, int64_t rows, float inv_cols,
                                                                float eps, float* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float ms = 0.f, m2 = 0.f;
    for (int i = 0; i < nparts; ++i) {
        const float2 v = part[(int64_t)i * rows + row];
        ms += v.x;
        m2 += v.y;
    }
    const float mean = ms / (float)nparts;
    float dev = 0.f;
    for (int i = 0; i < nparts; ++i) {
        const float d = part[(int64_t)i * rows + row].x - mean;      // (second read: L2)
        dev += d * d;
    }
    const float rstd = rsqrtf((m2 + (float)ME_STATS_GROUP * dev) * inv_cols + eps);
    *reinterpret_cast<float2*>(out + row * 2) = float2{rstd, -rstd * mean};
}

// generic fallback: any C, scalar accesses, row re-read from cache instead of registers
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_generic_kernel(const void* __restrict__ x, int x_dt,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta,
                                                                    void* __restrict__ y, int y_dt,
                                                                    float* __restrict__ mean_out,
                                                                    float* __restrict__ rstd_out, int64_t rows,
                                                                    int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += load1_as_f32(x, x_dt, row * C + c);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = load1_as_f32(x, x_dt, row * C + c) - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    for (int c = lane; c < C; c += 64)
        store1_from_f32(y, y_dt, row * C + c, (load1_as_f32(x, x_dt, row * C + c) - mean) * rstd * gamma[c] + beta[c]);
}

// ---- backward.  Per row:  xhat = (x-mean)*rstd ; g = dy*gamma ;
//   dx = rstd * (g - mean_c(g) - xhat * mean_c(g*xhat)) [+ dres]
//   dgamma[c] = sum_rows dy*xhat ; dbeta[c] = sum_rows dy
// Each wave walks rows with a grid stride; a lane always owns the same columns, so the affine partial sums
// live in registers and are written once per wave to the workspace [n_waves, 2, C]; a second kernel folds
// them (deterministic, no atomics).

// The grid IS the residency: every block walks rows / gridDim.x of the problem, so the launch takes exactly as many blocks as the
// chip holds at once (occupancy query per instantiation: the bf16 C = 768 form needs 136 - 168 registers = three blocks per CU;
// round 3's fixed 1024 blocks ran as 1 + 1/3 rounds there, the last 256 alone on a third of the chip's waves).  LNB_BLOCKS is the
// upper bound the partial-sum workspace is sized for.
constexpr int LNB_BLOCKS = 1024;

// RES (the residual gradient is added: every call of the encoder's backward) is a TEMPLATE parameter: as a run-time `if (dres)`
// around its loads hipcc waited vmcnt(0) behind each of the VPL load groups -- three serial memory round trips per pair of rows
// instead of one (ISA: L L L L b L L W1 W0 per group), 4.1 TB/s.
template <int VPL, typename TX, typename TDY, bool RES>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const void* __restrict__ dy, int dy_dt,
                                                            const void* __restrict__ x, int x_dt,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const void* __restrict__ dres, int dres_dt,
                                                            void* __restrict__ dx, int dx_dt,
                                                            float* __restrict__ partial, int want_affine,
                                                            int64_t rows, int C) {
    extern __shared__ __attribute__((aligned(16))) float lds_red[];     // [4 waves][2][C] when want_affine
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gw = blockIdx.x * (LN_THREADS / 64) + w;
    f32x4 g[VPL], dg[VPL], db[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        g[i] = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * 4);
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // two rows in flight per wave; every load of both rows (x, dy and the residual gradient) is issued up front
    const int64_t nwaves = (int64_t)gridDim.x * (LN_THREADS / 64);
    for (int64_t row0 = gw; row0 < rows; row0 += 2 * nwaves) {
        const int64_t row1 = row0 + nwaves;
        const bool has1 = row1 < rows;
        const int64_t r1 = has1 ? row1 : row0;
        const float mu0 = mean[row0], rs0 = rstd[row0], mu1 = mean[r1], rs1 = rstd[r1];
        f32x4 xh0[VPL], d0[VPL], xh1[VPL], d1[VPL], q0[VPL], q1[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int64_t c = (int64_t)(lane + 64 * i) * 4;
            xh0[i] = ld4<TX>(x, row0 * C + c);
            d0[i] = ld4<TDY>(dy, row0 * C + c);
            xh1[i] = ld4<TX>(x, r1 * C + c);
            d1[i] = ld4<TDY>(dy, r1 * C + c);
            if (RES) {
                q0[i] = ld4<TX>(dres, row0 * C + c);
                q1[i] = ld4<TX>(dres, r1 * C + c);
            } else {
                q0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                q1[i] = q0[i];
            }
        }
        float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
        const float w1 = has1 ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh0[i][e] = (xh0[i][e] - mu0) * rs0;
                xh1[i][e] = (xh1[i][e] - mu1) * rs1;
                const float ge0 = d0[i][e] * g[i][e], ge1 = d1[i][e] * g[i][e];
                a0 += ge0; b0 += ge0 * xh0[i][e];
                a1 += ge1; b1 += ge1 * xh1[i][e];
                dg[i][e] += d0[i][e] * xh0[i][e] + w1 * d1[i][e] * xh1[i][e];
                db[i][e] += d0[i][e] + w1 * d1[i][e];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a0 += __shfl_xor(a0, o, 64); b0 += __shfl_xor(b0, o, 64);
            a1 += __shfl_xor(a1, o, 64); b1 += __shfl_xor(b1, o, 64);
        }
        const float inv = 1.0f / (float)C;
        a0 *= inv; b0 *= inv; a1 *= inv; b1 *= inv;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int64_t c = (int64_t)(lane + 64 * i) * 4;
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] = rs0 * (d0[i][e] * g[i][e] - a0 - xh0[i][e] * b0) + q0[i][e];
                o1[e] = rs1 * (d1[i][e] * g[i][e] - a1 - xh1[i][e] * b1) + q1[i][e];
            }
            st4<TX>(dx, row0 * C + c, o0);
            if (has1) st4<TX>(dx, row1 * C + c, o1);
        }
    }
    if (want_affine) {     // deterministic block fold: waves park their partials in LDS, then columns are summed in wave order
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (lane + 64 * i) * 4;
            *reinterpret_cast<f32x4*>(lds_red + (w * 2 + 0) * C + c) = dg[i];
            *reinterpret_cast<f32x4*>(lds_red + (w * 2 + 1) * C + c) = db[i];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * C; c += LN_THREADS) {
            const int which = c / C, col = c % C;
            const float v = (lds_red[(0 * 2 + which) * C + col] + lds_red[(1 * 2 + which) * C + col]) +
                            (lds_red[(2 * 2 + which) * C + col] + lds_red[(3 * 2 + which) * C + col]);
            partial[((int64_t)blockIdx.x * 2 + which) * C + col] = v;
        }
    }
}

// fold the [nrows][2][C] dgamma / dbeta partials of up to LN_FOLD_SETS LayerNorm backward launches in ONE launch (me_block_bwd folds
// both LayerNorms of a block together at its end: round 3 ran two fold stages per LayerNorm = 4 tiny launches per block): grid (C / 64, sets), fixed-order sums.  Deterministic.  (C % 4 == 0: only the vector path of me_layernorm_bwd, C % 256 == 0, uses it.)
__global__ __launch_bounds__(1024) void ln_bwd_fold_sets_kernel(const me_ln_fold_batch fb) {
    __shared__ float sh[2][64][65];
    const me_ln_fold_set st = fb.set[blockIdx.y];
    const int C = fb.cols;
    // 16 lanes x 4 columns = the block's 64 columns of one partial row (256 contiguous bytes); 64 row slots per block, slot s takes
    // rows s, s + 64, ...: 12 x 2 sixteen-byte loads per thread at 768 partial rows, all independent
    const int cq = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + 4 * cq;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        for (int r = rs; r < st.nrows; r += 64) {
            a += *reinterpret_cast<const f32x4*>(st.partial + ((int64_t)r * 2 + 0) * C + c);
            b += *reinterpret_cast<const f32x4*>(st.partial + ((int64_t)r * 2 + 1) * C + c);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[0][rs][4 * cq + e] = a[e]; sh[1][rs][4 * cq + e] = b[e]; }
    __syncthreads();
    // 128 threads finish: thread (which, column) adds the 64 row slots in a fixed order
    if (threadIdx.x < 128) {
        const int w = threadIdx.x >> 6, col = threadIdx.x & 63, cc = blockIdx.x * 64 + col;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 64; i += 4)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += sh[w][i + e][col];
        const float tot = (t[0] + t[1]) + (t[2] + t[3]);
        if (cc < C) {
            float* out = w ? st.dbeta : st.dgamma;
            out[cc] = st.accumulate ? out[cc] + tot : tot;
        }
    }
}

// generic fallbacks (any C): scalar accesses; dx per row, affine grads by a column-walk kernel.
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_generic_dx_kernel(const void* __restrict__ dy, int dy_dt,
                                                                       const void* __restrict__ x, int x_dt,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd,
                                                                       const float* __restrict__ gamma,
                                                                       const void* __restrict__ dres, int dres_dt,
                                                                       void* __restrict__ dx, int dx_dt,
                                                                       int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float xh = (load1_as_f32(x, x_dt, row * C + c) - mu) * rs;
        const float ge = load1_as_f32(dy, dy_dt, row * C + c) * gamma[c];
        s1 += ge;
        s2 += ge * xh;
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (load1_as_f32(x, x_dt, row * C + c) - mu) * rs;
        const float ge = load1_as_f32(dy, dy_dt, row * C + c) * gamma[c];
        float o = rs * (ge - m1 - xh * m2);
        if (dres) o += load1_as_f32(dres, dres_dt, row * C + c);
        store1_from_f32(dx, dx_dt, row * C + c, o);
    }
}

__global__ __launch_bounds__(256) void ln_bwd_generic_affine_kernel(const void* __restrict__ dy, int dy_dt,
                                                                    const void* __restrict__ x, int x_dt,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd,
                                                                    float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, int64_t rows, int C,
                                                                    int accumulate) {
    __shared__ float sh[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a = 0.f, b = 0.f;
    if (c < C) {
        for (int64_t r = w; r < rows; r += 4) {
            const float d = load1_as_f32(dy, dy_dt, r * C + c);
            a += d * (load1_as_f32(x, x_dt, r * C + c) - mean[r]) * rstd[r];
            b += d;
        }
    }
    sh[0][w][lane] = a;
    sh[1][w][lane] = b;
    __syncthreads();
    if (w == 0 && c < C) {
        a = (sh[0][0][lane] + sh[0][1][lane]) + (sh[0][2][lane] + sh[0][3][lane]);
        b = (sh[1][0][lane] + sh[1][1][lane]) + (sh[1][2][lane] + sh[1][3][lane]);
        if (accumulate) {
            dgamma[c] += a;
            dbeta[c] += b;
        } else {
            dgamma[c] = a;
            dbeta[c] = b;
        }
    }
}

}  // namespace

extern "C" int me_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                                int y_dtype, float* mean, float* rstd, int64_t rows, int cols, float eps,
                                void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(x && y && gamma && beta, "me_layernorm_fwd: null pointer");
    ME_CHECK_ARG(me_dtype_ok(x_dtype) && me_out_dtype_ok(y_dtype), "me_layernorm_fwd: bad dtype");
    ME_CHECK_ARG(rows >= 0 && cols > 0, "me_layernorm_fwd: bad shape");
    // (the split output is what the fp32-accurate Linear behind it reads: fp32 tokens, vector path only)
    ME_CHECK_ARG(y_dtype != ME_BF16X3 || (x_dtype == ME_F32 && cols % 256 == 0 && cols / 256 <= LN_MAX_VEC),
                 "me_layernorm_fwd: ME_BF16X3 output needs fp32 input and cols = 256 * k <= %d", 256 * LN_MAX_VEC);
    if (rows == 0) return ME_OK;
    ProfScope prof(ME_PROF_LN_FWD, x_dtype, rows, cols, 0, stream);
    const unsigned nblk = (unsigned)((rows + 3) / 4);
#define LN_FWD_LAUNCH(V, TX, TY)                                                                               \
    hipLaunchKernelGGL((ln_fwd_kernel<V, TX, TY>), dim3((unsigned)((rows + 4 * ln_rows_per_wave(V) - 1) / (4 * ln_rows_per_wave(V)))), dim3(LN_THREADS), 0, stream, x, gamma, beta, y, mean, rstd,  \
                       rows, cols, eps)
#define LN_FWD_CASE(V)                                                                                         \
    case V:                                                                                                    \
        if (y_dtype == ME_BF16X3) LN_FWD_LAUNCH(V, float, me_split3_tag);                                      \
        else if (x_dtype == ME_BF16 && y_dtype == ME_BF16) LN_FWD_LAUNCH(V, bf16_t, bf16_t);                   \
        else if (x_dtype == ME_BF16) LN_FWD_LAUNCH(V, bf16_t, float);                                          \
        else if (y_dtype == ME_BF16) LN_FWD_LAUNCH(V, float, bf16_t);                                          \
        else LN_FWD_LAUNCH(V, float, float);                                                                   \
        break;
    if (cols % 256 == 0 && cols / 256 <= LN_MAX_VEC) {
        switch (cols / 256) {
            LN_FWD_CASE(1) LN_FWD_CASE(2) LN_FWD_CASE(3) LN_FWD_CASE(4) LN_FWD_CASE(5) LN_FWD_CASE(6)
            LN_FWD_CASE(7) LN_FWD_CASE(8) LN_FWD_CASE(9) LN_FWD_CASE(10) LN_FWD_CASE(11) LN_FWD_CASE(12)
            LN_FWD_CASE(13) LN_FWD_CASE(14) LN_FWD_CASE(15) LN_FWD_CASE(16)
        }
    } else {
        hipLaunchKernelGGL(ln_fwd_generic_kernel, dim3(nblk), dim3(LN_THREADS), 0, stream, x, x_dtype, gamma, beta,
                           y, y_dtype, mean, rstd, rows, cols, eps);
    }
#undef LN_FWD_CASE
#undef LN_FWD_LAUNCH
    ME_CHECK_LAUNCH("me_layernorm_fwd");
    return ME_OK;
}

extern "C" int me_row_stats(const void* x, int x_dtype, float* out, int64_t rows, int cols, float eps, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(x && out, "me_row_stats: null pointer");
    ME_CHECK_ARG(me_dtype_ok(x_dtype), "me_row_stats: bad dtype");
    ME_CHECK_ARG(rows >= 0 && cols > 0, "me_row_stats: bad shape");
    if (rows == 0) return ME_OK;
    ProfScope prof(ME_PROF_ROW_STATS, x_dtype, rows, cols, 0, stream);
    const unsigned nblk = (unsigned)((rows + 3) / 4);
#define LN_ST_CASE(V)                                                                                                          \
    case V:                                                                                                                    \
        if (x_dtype == ME_BF16) hipLaunchKernelGGL((ln_stats_kernel<V, bf16_t>), dim3((unsigned)((rows + 4 * ln_rows_per_wave(V) - 1) / (4 * ln_rows_per_wave(V)))), dim3(LN_THREADS), 0, stream, x, out, rows, cols, eps); \
        else hipLaunchKernelGGL((ln_stats_kernel<V, float>), dim3((unsigned)((rows + 4 * ln_rows_per_wave(V) - 1) / (4 * ln_rows_per_wave(V)))), dim3(LN_THREADS), 0, stream, x, out, rows, cols, eps);   \
        break;
    if (cols % 256 == 0 && cols / 256 <= 8) {
        switch (cols / 256) { LN_ST_CASE(1) LN_ST_CASE(2) LN_ST_CASE(3) LN_ST_CASE(4) LN_ST_CASE(5) LN_ST_CASE(6) LN_ST_CASE(7) LN_ST_CASE(8) }
    } else {
        hipLaunchKernelGGL(ln_stats_generic_kernel, dim3(nblk), dim3(LN_THREADS), 0, stream, x, x_dtype, out, rows, cols, eps);
    }
#undef LN_ST_CASE
    ME_CHECK_LAUNCH("me_row_stats");
    return ME_OK;
}

extern "C" size_t me_row_stats_partial_bytes(int64_t rows, int cols) {
    return rows > 0 && cols > 0 ? (size_t)(cols / ME_STATS_GROUP) * (size_t)rows * 2 * sizeof(float) : 0;
}

extern "C" int me_row_stats_combine(const float* partials, int64_t rows, int cols, float eps, float* out, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(partials && out, "me_row_stats_combine: null pointer");
    ME_CHECK_ARG(rows >= 0 && cols > 0 && cols % ME_STATS_GROUP == 0, "me_row_stats_combine: cols must be a positive multiple of 256");
    ME_CHECK_ARG((uintptr_t)partials % 8 == 0 && (uintptr_t)out % 8 == 0, "me_row_stats_combine: 8-byte aligned buffers");
    if (rows == 0) return ME_OK;
    ProfScope prof(ME_PROF_ROW_STATS, ME_F32, rows, cols, 1, stream);      // (K = 1: the combine pass; K = 0: me_row_stats over the tokens)
    hipLaunchKernelGGL(row_stats_combine_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(partials), cols / ME_STATS_GROUP, rows, 1.0f / (float)cols, eps, out);
    ME_CHECK_LAUNCH("me_row_stats_combine");
    return ME_OK;
}

// blocks of one ln_bwd_kernel instantiation the device holds at once (cached per instantiation, LDS size class AND device: a
// process may drive GPUs with different CU counts; the entries are written once each, under a lock)
template <auto KERNEL>
static int ln_bwd_resident_blocks(size_t lds_bytes) {
    static std::atomic<int> cached[64][2] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool cacheable = dev >= 0 && dev < 64;
    if (cacheable) {
        const int c = cached[dev][lds_bytes ? 1 : 0].load(std::memory_order_acquire);
        if (c) return c;
    }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, LN_THREADS, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const int c = per_cu * cus < LNB_BLOCKS ? per_cu * cus : LNB_BLOCKS;
    if (cacheable) cached[dev][lds_bytes ? 1 : 0].store(c, std::memory_order_release);      // (racing writers store the same value)
    return c;
}

extern "C" size_t me_layernorm_bwd_workspace(int cols) {
    return (size_t)LNB_BLOCKS * 2 * (size_t)cols * sizeof(float);
}

int me_ln_bwd_fold_sets(const me_ln_fold_set* sets, int n, int cols, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    me_ln_fold_batch fb;
    fb.cols = cols;
    int k = 0;
    for (int i = 0; i < n; ++i)
        if (sets[i].partial) fb.set[k++] = sets[i];          // (a set without partials was folded by its own launch: generic path)
    if (k == 0) return ME_OK;
    hipLaunchKernelGGL(ln_bwd_fold_sets_kernel, dim3((cols + 63) / 64, k), dim3(1024), 0, stream, fb);
    ME_CHECK_LAUNCH("me_layernorm_bwd(fold)");
    return ME_OK;
}

extern "C" int me_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                const float* rstd, const float* gamma, const void* dres, int dres_dtype, void* dx,
                                int dx_dtype, float* dgamma, float* dbeta, int accumulate_affine, int64_t rows,
                                int cols, void* workspace, void* stream_) {
    return me_ln_bwd_deferred(dy, dy_dtype, x, x_dtype, mean, rstd, gamma, dres, dres_dtype, dx, dx_dtype, dgamma, dbeta, accumulate_affine,
                              rows, cols, workspace, stream_, nullptr);
}

// defer != null: the dgamma / dbeta partials stay in `workspace` and *defer describes them for a later me_ln_bwd_fold_sets()
int me_ln_bwd_deferred(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd, const float* gamma,
                       const void* dres, int dres_dtype, void* dx, int dx_dtype, float* dgamma, float* dbeta, int accumulate_affine,
                       int64_t rows, int cols, void* workspace, void* stream_, me_ln_fold_set* defer) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (defer) defer->partial = nullptr;
    ProfScope prof(ME_PROF_LN_BWD, x_dtype, rows, cols, 0, stream);
    ME_CHECK_ARG(dy && x && mean && rstd && gamma && dx, "me_layernorm_bwd: null pointer");
    ME_CHECK_ARG(me_dtype_ok(dy_dtype) && me_dtype_ok(x_dtype) && me_dtype_ok(dx_dtype), "me_layernorm_bwd: bad dtype");
    ME_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "me_layernorm_bwd: dgamma/dbeta must both be given or both NULL");
    const int want_affine = dgamma != nullptr;
    ME_CHECK_ARG(!want_affine || workspace, "me_layernorm_bwd: workspace required for dgamma/dbeta");
    if (rows == 0) return ME_OK;
    const bool same_stream_dtype = dx_dtype == x_dtype && (dres == nullptr || dres_dtype == x_dtype);
    if (!(cols % 256 == 0 && cols / 256 <= 8) || !same_stream_dtype) {
        hipLaunchKernelGGL(ln_bwd_generic_dx_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(LN_THREADS), 0, stream, dy,
                           dy_dtype, x, x_dtype, mean, rstd, gamma, dres, dres_dtype, dx, dx_dtype, rows, cols);
        ME_CHECK_LAUNCH("me_layernorm_bwd(generic dx)");
        if (want_affine) {
            hipLaunchKernelGGL(ln_bwd_generic_affine_kernel, dim3((cols + 63) / 64), dim3(256), 0, stream, dy, dy_dtype,
                               x, x_dtype, mean, rstd, dgamma, dbeta, rows, cols, accumulate_affine);
            ME_CHECK_LAUNCH("me_layernorm_bwd(generic affine)");
        }
        return ME_OK;
    }
    float* partial = reinterpret_cast<float*>(workspace);
    const size_t lds_bytes = want_affine ? (size_t)4 * 2 * cols * sizeof(float) : 0;
    int nblocks = LNB_BLOCKS;
#define LN_BWD_LAUNCH(V, TX, TDY)                                                                                 \
    do {                                                                                                          \
        if (dres) {                                                                                               \
            nblocks = ln_bwd_resident_blocks<&ln_bwd_kernel<V, TX, TDY, true>>(lds_bytes);                          \
            hipLaunchKernelGGL((ln_bwd_kernel<V, TX, TDY, true>), dim3(nblocks), dim3(LN_THREADS), lds_bytes, stream, dy, dy_dtype, \
                               x, x_dtype, mean, rstd, gamma, dres, dres_dtype, dx, dx_dtype, partial, want_affine, rows, cols);    \
        } else {                                                                                                  \
            nblocks = ln_bwd_resident_blocks<&ln_bwd_kernel<V, TX, TDY, false>>(lds_bytes);                         \
            hipLaunchKernelGGL((ln_bwd_kernel<V, TX, TDY, false>), dim3(nblocks), dim3(LN_THREADS), lds_bytes, stream, dy, dy_dtype, \
                               x, x_dtype, mean, rstd, gamma, dres, dres_dtype, dx, dx_dtype, partial, want_affine, rows, cols);    \
        }                                                                                                         \
    } while (0)
#define LN_BWD_CASE(V)                                                                                            \
    case V:                                                                                                       \
        if (x_dtype == ME_BF16 && dy_dtype == ME_BF16) LN_BWD_LAUNCH(V, bf16_t, bf16_t);                          \
        else if (x_dtype == ME_BF16) LN_BWD_LAUNCH(V, bf16_t, float);                                             \
        else if (dy_dtype == ME_BF16) LN_BWD_LAUNCH(V, float, bf16_t);                                            \
        else LN_BWD_LAUNCH(V, float, float);                                                                      \
        break;
    switch (cols / 256) {
        LN_BWD_CASE(1) LN_BWD_CASE(2) LN_BWD_CASE(3) LN_BWD_CASE(4) LN_BWD_CASE(5) LN_BWD_CASE(6) LN_BWD_CASE(7)
        LN_BWD_CASE(8)
    }
#undef LN_BWD_CASE
    ME_CHECK_LAUNCH("me_layernorm_bwd");
    if (want_affine) {
        me_ln_fold_set st;
        st.partial = partial; st.nrows = nblocks; st.dgamma = dgamma; st.dbeta = dbeta; st.accumulate = accumulate_affine;
        if (defer) *defer = st;
        else return me_ln_bwd_fold_sets(&st, 1, cols, stream_);
    }
    return ME_OK;
}
