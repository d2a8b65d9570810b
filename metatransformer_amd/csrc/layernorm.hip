// layernorm.hip -- LayerNorm forward / backward over the channel dim of [rows, C] tokens.
//
// Replaces nn.LayerNorm(C, eps) (Block.norm1 / norm2, PointCloud/openpoints/models/layers/attention.py:46,50
// -> norm.py:65).  HBM-bound: one wavefront per token row, 8/16-byte vector loads, the row is held in
// registers between the statistics pass and the normalise pass (one HBM read, one write), statistics by
// wavefront xor-shuffles in fp32 with the two-pass (mean, then centred sum of squares) formulation.
#include "common.h"

namespace {

constexpr int LN_THREADS = 256;           // 4 rows per block
constexpr int LN_MAX_VEC = 16;            // supports C up to 4*64*16 = 4096 on the vector path

// VPL = number of 4-element vectors per lane (C = 256 * VPL on the fast path)
template <int VPL>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const void* __restrict__ x, int x_dt,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, void* __restrict__ y,
                                                            int y_dt, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int64_t rows, int C,
                                                            float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = load4_as_f32(x, x_dt, row * C + (int64_t)(lane + 64 * i) * 4);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            q += d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (lane + 64 * i) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        store4_from_f32(y, y_dt, row * C + c, o);
    }
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
constexpr int LNB_BLOCKS = 512;
constexpr int LNB_WAVES = LNB_BLOCKS * (LN_THREADS / 64);

template <int VPL>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const void* __restrict__ dy, int dy_dt,
                                                            const void* __restrict__ x, int x_dt,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const void* __restrict__ dres, int dres_dt,
                                                            void* __restrict__ dx, int dx_dt,
                                                            float* __restrict__ partial, int want_affine,
                                                            int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
    f32x4 g[VPL], dg[VPL], db[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        g[i] = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * 4);
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t row = gw; row < rows; row += LNB_WAVES) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[VPL], d[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int64_t idx = row * C + (int64_t)(lane + 64 * i) * 4;
            const f32x4 xv = load4_as_f32(x, x_dt, idx);
            d[i] = load4_as_f32(dy, dy_dt, idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = (xv[e] - mu) * rs;
                const float ge = d[i][e] * g[i][e];
                s1 += ge;
                s2 += ge * xh[i][e];
                dg[i][e] += d[i][e] * xh[i][e];
                db[i][e] += d[i][e];
            }
        }
        const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int64_t idx = row * C + (int64_t)(lane + 64 * i) * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (d[i][e] * g[i][e] - m1 - xh[i][e] * m2);
            if (dres) o += load4_as_f32(dres, dres_dt, idx);
            store4_from_f32(dx, dx_dt, idx, o);
        }
    }
    if (want_affine) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (lane + 64 * i) * 4;
            *reinterpret_cast<f32x4*>(partial + ((int64_t)gw * 2 + 0) * C + c) = dg[i];
            *reinterpret_cast<f32x4*>(partial + ((int64_t)gw * 2 + 1) * C + c) = db[i];
        }
    }
}

__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partial,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int C, int accumulate) {
    // one block per 64 columns; 4 waves split the partial rows, lanes = columns
    __shared__ float sh[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a = 0.f, b = 0.f;
    if (c < C) {
        for (int r = w; r < LNB_WAVES; r += 4) {
            a += partial[((int64_t)r * 2 + 0) * C + c];
            b += partial[((int64_t)r * 2 + 1) * C + c];
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
    ME_CHECK_ARG(me_dtype_ok(x_dtype) && me_dtype_ok(y_dtype), "me_layernorm_fwd: bad dtype");
    ME_CHECK_ARG(rows >= 0 && cols > 0, "me_layernorm_fwd: bad shape");
    if (rows == 0) return ME_OK;
    const unsigned nblk = (unsigned)((rows + 3) / 4);
#define LN_FWD_CASE(V)                                                                                         \
    case V:                                                                                                    \
        hipLaunchKernelGGL((ln_fwd_kernel<V>), dim3(nblk), dim3(LN_THREADS), 0, stream, x, x_dtype, gamma,      \
                           beta, y, y_dtype, mean, rstd, rows, cols, eps);                                     \
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
    ME_CHECK_LAUNCH("me_layernorm_fwd");
    return ME_OK;
}

extern "C" size_t me_layernorm_bwd_workspace(int cols) { return (size_t)LNB_WAVES * 2 * (size_t)cols * sizeof(float); }

extern "C" int me_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                const float* rstd, const float* gamma, const void* dres, int dres_dtype, void* dx,
                                int dx_dtype, float* dgamma, float* dbeta, int accumulate_affine, int64_t rows,
                                int cols, void* workspace, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(dy && x && mean && rstd && gamma && dx, "me_layernorm_bwd: null pointer");
    ME_CHECK_ARG(me_dtype_ok(dy_dtype) && me_dtype_ok(x_dtype) && me_dtype_ok(dx_dtype), "me_layernorm_bwd: bad dtype");
    ME_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "me_layernorm_bwd: dgamma/dbeta must both be given or both NULL");
    const int want_affine = dgamma != nullptr;
    ME_CHECK_ARG(!want_affine || workspace, "me_layernorm_bwd: workspace required for dgamma/dbeta");
    if (rows == 0) return ME_OK;
    if (!(cols % 256 == 0 && cols / 256 <= 8)) {
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
#define LN_BWD_CASE(V)                                                                                          \
    case V:                                                                                                     \
        hipLaunchKernelGGL((ln_bwd_kernel<V>), dim3(LNB_BLOCKS), dim3(LN_THREADS), 0, stream, dy, dy_dtype, x,   \
                           x_dtype, mean, rstd, gamma, dres, dres_dtype, dx, dx_dtype, partial, want_affine,    \
                           rows, cols);                                                                         \
        break;
    switch (cols / 256) {
        LN_BWD_CASE(1) LN_BWD_CASE(2) LN_BWD_CASE(3) LN_BWD_CASE(4) LN_BWD_CASE(5) LN_BWD_CASE(6) LN_BWD_CASE(7)
        LN_BWD_CASE(8)
    }
#undef LN_BWD_CASE
    ME_CHECK_LAUNCH("me_layernorm_bwd");
    if (want_affine) {
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((cols + 63) / 64), dim3(256), 0, stream, partial, dgamma, dbeta,
                           cols, accumulate_affine);
        ME_CHECK_LAUNCH("me_layernorm_bwd(reduce)");
    }
    return ME_OK;
}
