// elementwise.hip -- HBM-bound helpers around the encoder: casts / weight repacks, bias-gradient column sums,
// pos-embed row adds, the Data2Seq patch gathers and the time-series embedding, fused AdamW.
// All of these are bandwidth-bound integer/index or element-wise work: coalesced 16-byte accesses, grid-stride
// loops, no LDS unless a transpose needs it.
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;
inline unsigned ew_blocks(int64_t work_items) {
    int64_t b = (work_items + EW_THREADS - 1) / EW_THREADS;
    const int64_t cap = 256 * 8;   // 256 CUs x 8 blocks, grid-stride the rest
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// fp16 exists at the boundary only (ME_F16: storage dtype of me_cast / me_transpose_cast -- `.half()` checkpoints and
// fp16-autocast call sites are converted to bf16 / fp32 on the way in and back on the way out; no kernel computes in it)
__device__ __forceinline__ float ld1_any(const void* base, int dt, int64_t i) {
    if (dt == ME_F16) return (float)reinterpret_cast<const _Float16*>(base)[i];
    return load1_as_f32(base, dt, i);
}
__device__ __forceinline__ void st1_any(void* base, int dt, int64_t i, float v) {
    if (dt == ME_F16) reinterpret_cast<_Float16*>(base)[i] = (_Float16)v;
    else store1_from_f32(base, dt, i, v);
}
__global__ __launch_bounds__(EW_THREADS) void cast_any_kernel(const void* __restrict__ src, int sdt, void* __restrict__ dst,
                                                              int ddt, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS)
        st1_any(dst, ddt, i, ld1_any(src, sdt, i));
}

__global__ __launch_bounds__(EW_THREADS) void cast_kernel(const void* __restrict__ src, int sdt, void* __restrict__ dst,
                                                          int ddt, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_THREADS)
        store4_from_f32(dst, ddt, i * 4, load4_as_f32(src, sdt, i * 4));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        store1_from_f32(dst, ddt, i, load1_as_f32(src, sdt, i));
    }
}

// fp32 [rows, cols] -> ME_BF16X3 [rows, 3 * cols]: one read of the fp32 values, three bf16 plane stores (6 bytes per element)
__global__ __launch_bounds__(EW_THREADS) void split3_kernel(const float* __restrict__ src, int64_t ld_src, uint16_t* __restrict__ dst,
                                                            int64_t rows, int64_t cols, int right_operand) {
    const int64_t q = cols / 4, n4 = rows * q;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t r = i / q, c = (i - r * q) * 4;
        store4_split3(dst + r * 3 * cols, cols, c, *reinterpret_cast<const f32x4*>(src + r * ld_src + c), right_operand != 0);
    }
}

// dst[c, r] = src[r, c]; 64x64 tiles through LDS (padded), coalesced on both sides
__global__ __launch_bounds__(256) void transpose_cast_kernel(const void* __restrict__ src, int sdt, void* __restrict__ dst,
                                                             int ddt, int64_t rows, int64_t cols) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? ld1_any(src, sdt, r * cols + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) st1_any(dst, ddt, c * rows + r, tile[tx][i]);
    }
}

// fp32 weights -> ME_BF16X3 right-operand planes [hi | hi | lo], of W itself (transposed = 0: dst [rows, 3 cols]) or of W^T
// (transposed = 1: dst [cols, 3 rows], the dgrad GEMMs' B operand), for up to ME_TC_BATCH matrices in one launch.  Same 64 x 64 tiles as
// the batched transpose; rows and cols multiples of 4.
__global__ __launch_bounds__(256) void split3_batched_kernel(const me_tc_batch b, int transposed, int right_operand) {
    __shared__ float tile[64][65];
    int64_t t = blockIdx.x;
    int k = 0;
    int64_t tx_tiles = 0;
    for (; k < b.n; ++k) {
        tx_tiles = (b.item[k].cols + 63) / 64;
        const int64_t nt = tx_tiles * ((b.item[k].rows + 63) / 64);
        if (t < nt) break;
        t -= nt;
    }
    if (k >= b.n) return;
    const float* src = reinterpret_cast<const float*>(b.item[k].src);
    uint16_t* dst = reinterpret_cast<uint16_t*>(b.item[k].dst);
    const int64_t rows = b.item[k].rows, cols = b.item[k].cols;
    const int64_t r0 = (t / tx_tiles) * 64, c0 = (t % tx_tiles) * 64;
    const int q = threadIdx.x & 15, p = threadIdx.x >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int i = p + 16 * pass;
        const int64_t r = r0 + i, c = c0 + 4 * q;
        if (r < rows && c < cols) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * cols + c);
            if (!transposed) {
                store4_split3(dst + r * 3 * cols, cols, c, v, right_operand != 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[i][4 * q + e] = v[e];
            }
        }
    }
    if (!transposed) return;          // (block-uniform)
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int i = p + 16 * pass;
        const int64_t c = c0 + i, r = r0 + 4 * q;
        if (c < cols && r < rows)
            store4_split3(dst + c * 3 * rows, rows, r, f32x4{tile[4 * q][i], tile[4 * q + 1][i], tile[4 * q + 2][i], tile[4 * q + 3][i]}, right_operand != 0);
    }
}

// batched form: the block finds its matrix by walking the (<= 48 entry) tile-count prefix
__global__ __launch_bounds__(256) void transpose_cast_batched_kernel(const me_tc_batch b) {
    __shared__ float tile[64][65];
    int64_t t = blockIdx.x;
    int k = 0;
    int64_t tx_tiles = 0;
    for (; k < b.n; ++k) {
        tx_tiles = (b.item[k].cols + 63) / 64;
        const int64_t nt = tx_tiles * ((b.item[k].rows + 63) / 64);
        if (t < nt) break;
        t -= nt;
    }
    if (k >= b.n) return;
    const void* src = b.item[k].src;
    void* dst = b.item[k].dst;
    const int64_t rows = b.item[k].rows, cols = b.item[k].cols;
    const int64_t r0 = (t / tx_tiles) * 64, c0 = (t % tx_tiles) * 64;
    const int ssz = b.src_dtype == ME_F32 ? 4 : 2, dsz = b.dst_dtype == ME_F32 ? 4 : 2;
    if (b.src_dtype != ME_F16 && b.dst_dtype != ME_F16 && rows % 4 == 0 && cols % 4 == 0 && (uintptr_t)src % (4 * ssz) == 0 && (uintptr_t)dst % (4 * dsz) == 0) {      // (block-uniform)
        // four elements per lane on both sides: 16 lanes read 64 columns of a row (128 / 256 contiguous bytes), 16 lanes write 64 rows
        // of a destination row; the 65-word tile rows keep both LDS passes conflict-free (bank = row + column)
        const int q = threadIdx.x & 15, p = threadIdx.x >> 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int i = p + 16 * pass;
            const int64_t r = r0 + i, c = c0 + 4 * q;
            const f32x4 v = (r < rows && c < cols) ? load4_as_f32(src, b.src_dtype, r * cols + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[i][4 * q + e] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int i = p + 16 * pass;
            const int64_t c = c0 + i, r = r0 + 4 * q;
            if (c < cols && r < rows) store4_from_f32(dst, b.dst_dtype, c * rows + r, f32x4{tile[4 * q][i], tile[4 * q + 1][i], tile[4 * q + 2][i], tile[4 * q + 3][i]});
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? ld1_any(src, b.src_dtype, r * cols + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) st1_any(dst, b.dst_dtype, c * rows + r, tile[tx][i]);
    }
}

__global__ __launch_bounds__(EW_THREADS) void add_rows_kernel(const void* __restrict__ x, int xdt,
                                                              const void* __restrict__ pos, int pdt, void* __restrict__ y,
                                                              int ydt, int64_t rows, int64_t pos_rows, int cols) {
    const int c4 = cols / 4;
    const int64_t total = rows * c4;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t r = i / c4;
        const int c = (int)(i % c4) * 4;
        f32x4 v = load4_as_f32(x, xdt, r * cols + c);
        v += load4_as_f32(pos, pdt, (r % pos_rows) * cols + c);
        store4_from_f32(y, ydt, r * cols + c, v);
    }
}

// ---- column sums (bias gradients): stage 1: grid (col blocks of 256, row splits); each thread owns 4 columns? no:
// lanes own single columns (coalesced 2/4-byte reads across the wave are fine for this size); 4 waves x splits rows.
constexpr int CS_SPLITS = 128;
// vector path (cols % 4 == 0): a lane owns 4 consecutive columns (8/16-byte loads), a wave covers 256 columns of a
// row, the 4 waves of a block and blockIdx.y split the rows; 4 rows in flight per wave.
// (dtype is a template parameter: a runtime switch around each load would put a vmcnt(0) at every branch join and
// serialise the four rows in flight)
template <typename T> __device__ __forceinline__ f32x4 cs_ld4(const void* base, int64_t idx);
template <> __device__ __forceinline__ f32x4 cs_ld4<float>(const void* base, int64_t idx) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
}
template <> __device__ __forceinline__ f32x4 cs_ld4<bf16_t>(const void* base, int64_t idx) {
    const u32x2 raw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(base) + idx);
    return f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u), __uint_as_float(raw[1] << 16),
                 __uint_as_float(raw[1] & 0xffff0000u)};
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const void* __restrict__ x, int64_t ldx,
                                                                 int64_t rows, int64_t cols, float* __restrict__ partial) {
    __shared__ f32x4 sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const int64_t rows_per = (rows + CS_SPLITS - 1) / CS_SPLITS;
    const int64_t rb = (int64_t)blockIdx.y * rows_per;
    const int64_t re = rb + rows_per < rows ? rb + rows_per : rows;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (c < cols) {
        int64_t r = rb + w;
        for (; r + 28 < re; r += 32) {
            const f32x4 v0 = cs_ld4<T>(x, r * ldx + c), v1 = cs_ld4<T>(x, (r + 4) * ldx + c);
            const f32x4 v2 = cs_ld4<T>(x, (r + 8) * ldx + c), v3 = cs_ld4<T>(x, (r + 12) * ldx + c);
            const f32x4 v4 = cs_ld4<T>(x, (r + 16) * ldx + c), v5 = cs_ld4<T>(x, (r + 20) * ldx + c);
            const f32x4 v6 = cs_ld4<T>(x, (r + 24) * ldx + c), v7 = cs_ld4<T>(x, (r + 28) * ldx + c);
            a0 += v0 + v4; a1 += v1 + v5; a2 += v2 + v6; a3 += v3 + v7;
        }
        for (; r < re; r += 4) a0 += cs_ld4<T>(x, r * ldx + c);
    }
    sh[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && c < cols)
        *reinterpret_cast<f32x4*>(partial + (int64_t)blockIdx.y * cols + c) = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* __restrict__ x, int xdt, int64_t ldx, int64_t rows,
                                                             int64_t cols, float* __restrict__ partial) {
    // scalar fallback.  block: 64 columns x 4 row-lanes; blockIdx.y = row split
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int64_t rows_per = (rows + CS_SPLITS - 1) / CS_SPLITS;
    const int64_t rb = (int64_t)blockIdx.y * rows_per;
    const int64_t re = rb + rows_per < rows ? rb + rows_per : rows;
    float a = 0.f;
    if (c < cols)
        for (int64_t r = rb + w; r < re; r += 4) a += load1_as_f32(x, xdt, r * ldx + c);
    sh[w][lane] = a;
    __syncthreads();
    if (w == 0 && c < cols)
        partial[(int64_t)blockIdx.y * cols + c] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int64_t cols,
                                                           float* __restrict__ out, int accumulate) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float a = 0.f;
    for (int s = 0; s < CS_SPLITS; ++s) a += partial[(int64_t)s * cols + c];
    out[c] = accumulate ? out[c] + a : a;
}

// out[c] = sum_r x[r,c] * y[r,c]  (layer-scale gradients: d gamma = colsum(dy * branch)); same split / fold as colsum
__global__ __launch_bounds__(256) void colsum_mul_partial_kernel(const void* __restrict__ x, int xdt, int64_t ldx,
                                                                 const void* __restrict__ y, int ydt, int64_t ldy,
                                                                 int64_t rows, int64_t cols, float* __restrict__ partial) {
    __shared__ f32x4 sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const int64_t rows_per = (rows + CS_SPLITS - 1) / CS_SPLITS;
    const int64_t rb = (int64_t)blockIdx.y * rows_per;
    const int64_t re = rb + rows_per < rows ? rb + rows_per : rows;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < cols)
        for (int64_t r = rb + w; r < re; r += 4) a += load4_as_f32(x, xdt, r * ldx + c) * load4_as_f32(y, ydt, r * ldy + c);
    sh[w][lane] = a;
    __syncthreads();
    if (w == 0 && c < cols)
        *reinterpret_cast<f32x4*>(partial + (int64_t)blockIdx.y * cols + c) = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}


// ---- window partition / merge of token rows (windowed attention, Image/detection/.../base/vit.py:160-190).
// Tokens of one image are rows (y, x) of an H x W grid; windows of ws x ws tile the grid padded up to multiples of ws.
// partition: win[(b, wy, wx), (iy, ix)] = tok[b, wy*ws + iy, wx*ws + ix]  or 0 outside the grid (the reference zero-pads
// q, k, v AFTER the qkv Linear, so padded keys take part in the softmax with score 0 and value 0); merge is the inverse
// gather (padded rows are dropped).  Same-dtype 16-byte row-chunk copies; pure integer index arithmetic (bit-exact).
__global__ __launch_bounds__(EW_THREADS) void window_rows_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int B,
                                                                 int H, int W, int ws, int chunks, int merge) {
    const int gh = (H + ws - 1) / ws, gw = (W + ws - 1) / ws;
    const int64_t win_rows = (int64_t)B * gh * gw * ws * ws, tok_rows = (int64_t)B * H * W;
    const int64_t total = (merge ? tok_rows : win_rows) * chunks;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t r = i / chunks;
        const int c = (int)(i % chunks);
        if (merge) {
            const int x = (int)(r % W), y = (int)((r / W) % H);
            const int64_t b = r / ((int64_t)W * H);
            const int64_t wr = (((b * gh + y / ws) * gw + x / ws) * ws + y % ws) * ws + x % ws;
            dst[i] = src[wr * chunks + c];
        } else {
            const int ix = (int)(r % ws), iy = (int)((r / ws) % ws);
            const int64_t w = r / ((int64_t)ws * ws);
            const int wx = (int)(w % gw), wy = (int)((w / gw) % gh);
            const int64_t b = w / ((int64_t)gw * gh);
            const int y = wy * ws + iy, x = wx * ws + ix;
            dst[i] = (y < H && x < W) ? src[((b * H + y) * W + x) * chunks + c] : u32x4{0u, 0u, 0u, 0u};
        }
    }
}

// ---- patch gather: cols[(b, pt, py, px), (c, dt, dy, dx)] = x[b, c, pt*st+dt, py*sh+dy, px*sw+dx]
struct PatchGeom {
    int B, Cin, T, H, W, kt, kh, kw, st, sh, sw, gt, gh, gw;
};
__global__ __launch_bounds__(EW_THREADS) void patchify_kernel(const void* __restrict__ x, int xdt, void* __restrict__ out,
                                                              int odt, PatchGeom g) {
    const int64_t feat = (int64_t)g.Cin * g.kt * g.kh * g.kw;
    const int64_t ntok = (int64_t)g.B * g.gt * g.gh * g.gw;
    const int64_t total = ntok * feat;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t tok = i / feat;
        int64_t f = i % feat;
        const int dx = (int)(f % g.kw); f /= g.kw;
        const int dy = (int)(f % g.kh); f /= g.kh;
        const int dt = (int)(f % g.kt); f /= g.kt;
        const int c = (int)f;
        int64_t t = tok;
        const int px = (int)(t % g.gw); t /= g.gw;
        const int py = (int)(t % g.gh); t /= g.gh;
        const int pt = (int)(t % g.gt); t /= g.gt;
        const int b = (int)t;
        const int64_t src = ((((int64_t)b * g.Cin + c) * g.T + (pt * g.st + dt)) * g.H + (py * g.sh + dy)) * g.W + (px * g.sw + dx);
        store1_from_f32(out, odt, i, load1_as_f32(x, xdt, src));
    }
}
// The same gather four features at a time (kw % 4 == 0: four consecutive dx of one patch row are four consecutive pixels): one index
// decomposition per quad, one 8 / 16-byte store per thread -- consecutive threads write consecutive bytes of the gathered matrix -- and a
// vector load where the source quad is aligned (x, W and sw multiples of 4 elements: the image / tubelet embeds), four scalar loads
// otherwise (the spectrogram's stride 10).  The scalar kernel above: 225 us for 256 images (0.7 TB/s); this one: see DESIGN.md.
__global__ __launch_bounds__(EW_THREADS) void patchify4_kernel(const void* __restrict__ x, int xdt, void* __restrict__ out, int odt, PatchGeom g,
                                                               int aligned) {
    const int64_t feat4 = (int64_t)g.Cin * g.kt * g.kh * (g.kw / 4);
    const int64_t ntok = (int64_t)g.B * g.gt * g.gh * g.gw;
    const int64_t total = ntok * feat4;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t tok = i / feat4;
        int64_t f = i - tok * feat4;
        const int dx = (int)(f % (g.kw / 4)) * 4; f /= (g.kw / 4);
        const int dy = (int)(f % g.kh); f /= g.kh;
        const int dt = (int)(f % g.kt); f /= g.kt;
        const int c = (int)f;
        int64_t t = tok;
        const int px = (int)(t % g.gw); t /= g.gw;
        const int py = (int)(t % g.gh); t /= g.gh;
        const int pt = (int)(t % g.gt); t /= g.gt;
        const int b = (int)t;
        const int64_t src = ((((int64_t)b * g.Cin + c) * g.T + (pt * g.st + dt)) * g.H + (py * g.sh + dy)) * g.W + (px * g.sw + dx);
        f32x4 v;
        if (aligned) {
            v = load4_as_f32(x, xdt, src);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = load1_as_f32(x, xdt, src + e);
        }
        store4_from_f32(out, odt, i * 4, v);
    }
}
__global__ __launch_bounds__(EW_THREADS) void unpatchify_add_kernel(const void* __restrict__ dcols, int ddt,
                                                                    float* __restrict__ dx_out, PatchGeom g) {
    const int64_t feat = (int64_t)g.Cin * g.kt * g.kh * g.kw;
    const int64_t ntok = (int64_t)g.B * g.gt * g.gh * g.gw;
    const int64_t total = ntok * feat;
    const bool overlap = g.st < g.kt || g.sh < g.kh || g.sw < g.kw;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t tok = i / feat;
        int64_t f = i % feat;
        const int dx = (int)(f % g.kw); f /= g.kw;
        const int dy = (int)(f % g.kh); f /= g.kh;
        const int dt = (int)(f % g.kt); f /= g.kt;
        const int c = (int)f;
        int64_t t = tok;
        const int px = (int)(t % g.gw); t /= g.gw;
        const int py = (int)(t % g.gh); t /= g.gh;
        const int pt = (int)(t % g.gt); t /= g.gt;
        const int b = (int)t;
        const int64_t dst = ((((int64_t)b * g.Cin + c) * g.T + (pt * g.st + dt)) * g.H + (py * g.sh + dy)) * g.W + (px * g.sw + dx);
        const float v = load1_as_f32(dcols, ddt, i);
        if (overlap) atomicAdd(dx_out + dst, v);
        else dx_out[dst] += v;
    }
}

// ---- time-series DataEmbedding
constexpr int TS_MAX_MARK = 8;
struct TsTables {
    const float* tab[TS_MAX_MARK];
    int rows[TS_MAX_MARK];
};
__global__ __launch_bounds__(EW_THREADS) void ts_embed_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const int32_t* __restrict__ marks, int n_mark, TsTables tt,
                                                              const float* __restrict__ pe, void* __restrict__ out, int odt,
                                                              int B, int L, int cin, int C, int32_t* __restrict__ err) {
    // one block per (b, l) token; threads over channels.  Reference summation order (Data2Seq/Time_Series.py:93,123):
    // value + (hour + weekday + day + month [+ minute]) + pos, i.e. mark columns 3,2,1,0[,4].
    const int64_t tok = blockIdx.x;
    const int b = (int)(tok / L), l = (int)(tok % L);
    const int lm = (l - 1 + L) % L, lp = (l + 1) % L;
    const float* x0 = x + ((int64_t)b * L + lm) * cin;
    const float* x1 = x + ((int64_t)b * L + l) * cin;
    const float* x2 = x + ((int64_t)b * L + lp) * cin;
    for (int c = threadIdx.x; c < C; c += EW_THREADS) {
        const float* wc = w + (int64_t)c * cin * 3;
        float v = 0.f;
        for (int i = 0; i < cin; ++i) v += wc[i * 3 + 0] * x0[i];
        float v1 = 0.f;
        for (int i = 0; i < cin; ++i) v1 += wc[i * 3 + 1] * x1[i];
        float v2 = 0.f;
        for (int i = 0; i < cin; ++i) v2 += wc[i * 3 + 2] * x2[i];
        v = (v + v1) + v2;
        if (marks) {
            float t = 0.f;
            const int order[5] = {3, 2, 1, 0, 4};
            bool first = true;
            for (int k = 0; k < 5; ++k) {
                const int f = order[k];
                if (f >= n_mark) continue;
                int idx = marks[tok * n_mark + f];
                if (idx < 0 || idx >= tt.rows[f]) {
                    if (err) atomicExch(err, 1);
                    idx = 0;
                }
                const float e = tt.tab[f][(int64_t)idx * C + c];
                t = first ? e : t + e;
                first = false;
            }
            v += t;
        }
        if (pe) v += pe[(int64_t)l * C + c];
        store1_from_f32(out, odt, tok * C + c, v);
    }
}


// ---- dropout / drop-path with residual add.  Counter-based RNG (splitmix64 of seed + element index): the same
// (seed, index) regenerates the same mask in backward, nothing is stored.
__global__ __launch_bounds__(EW_THREADS) void dropout_add_kernel(const void* __restrict__ v, int vdt,
                                                                 const void* __restrict__ res, int rdt,
                                                                 void* __restrict__ out, int odt, int64_t rows, int cols,
                                                                 int64_t rows_per_sample, float p_drop, float p_path,
                                                                 uint64_t seed, const float* __restrict__ colscale) {
    const int c4 = cols / 4;
    const int64_t total = rows * c4;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const float inv_path = p_path > 0.f ? 1.0f / (1.0f - p_path) : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t r = i / c4;
        const int c = (int)(i % c4) * 4;
        float rowscale = 1.0f;
        if (p_path > 0.f) {
            const uint64_t sample = (uint64_t)(r / rows_per_sample);
            rowscale = u01_hash(seed ^ 0xD1B54A32D192ED03ull, sample) >= p_path ? inv_path : 0.0f;
        }
        f32x4 x = load4_as_f32(v, vdt, r * cols + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float k = rowscale;
            if (p_drop > 0.f) k = u01_hash(seed, (uint64_t)(r * cols + c + e)) >= p_drop ? k * inv_keep : 0.0f;
            x[e] *= k;
        }
        if (colscale) x *= *reinterpret_cast<const f32x4*>(colscale + c);
        if (res) x += load4_as_f32(res, rdt, r * cols + c);
        store4_from_f32(out, odt, r * cols + c, x);
    }
}

__device__ __forceinline__ float adamw_update(float& pi, float gi, float& mi, float& vi, float lr, float b1, float b2, float eps, float wd,
                                              float bc1, float bc2_sqrt) {
    pi *= (1.0f - lr * wd);
    mi = b1 * mi + (1.0f - b1) * gi;
    vi = b2 * vi + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    return pi;
}
// VEC: four elements per lane (16-byte accesses on the four fp32 streams, 8 bytes on the bf16 mirror); the launcher takes it when the
// slice's addresses allow it, the n % 4 tail goes through block 0's first lanes.  Same arithmetic per element either way.
template <bool VEC>
__global__ __launch_bounds__(EW_THREADS) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                           float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                           float gscale, bf16_t* __restrict__ mirror) {
    if (VEC) {
        const int64_t n4 = n / 4;
        for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_THREADS) {
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + i * 4), mv = *reinterpret_cast<const f32x4*>(m + i * 4),
                  vv = *reinterpret_cast<const f32x4*>(v + i * 4);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pi = pv[e], mi = mv[e], vi = vv[e];
                adamw_update(pi, gv[e] * gscale, mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
                pv[e] = pi; mv[e] = mi; vv[e] = vi;
            }
            *reinterpret_cast<f32x4*>(m + i * 4) = mv;
            *reinterpret_cast<f32x4*>(v + i * 4) = vv;
            *reinterpret_cast<f32x4*>(p + i * 4) = pv;
            if (mirror) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)pv[e];
                *reinterpret_cast<bf16x4*>(mirror + i * 4) = o;
            }
        }
        if (blockIdx.x != 0 || (int64_t)threadIdx.x >= n - n4 * 4) return;
        const int64_t i = n4 * 4 + threadIdx.x;
        float pi = p[i], mi = m[i], vi = v[i];
        adamw_update(pi, g[i] * gscale, mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (mirror) mirror[i] = (bf16_t)pi;
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        float pi = p[i], mi = m[i], vi = v[i];
        adamw_update(pi, g[i] * gscale, mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
        if (mirror) mirror[i] = (bf16_t)pi;
    }
}

// ---- fine-tune recipe of the fused optimizer: what the reference's training loops wrap around AdamW --
// layer-wise lr decay / no-decay parameter groups (Video/optim_factory.py:28-41, 56-95;
// Image/segmentation/mmcv_custom/layer_decay_optimizer_constructor.py:17-41), GradScaler.unscale_ + clip_grad_norm_ and the
// skipped step on a non-finite gradient (Video/utils.py:376-404) -- on the flat bucket, with the decisions taken ON THE
// DEVICE: no host round trip between backward and the optimizer step.
//   grad_stats_kernel + grad_stats_fold_kernel : L2 norm and number of non-finite values of the bucket (deterministic
//                                                two-level sum of squares in double)
//   adamw_prepare_kernel                       : one thread: total norm, clip coefficient, found-inf, the step counter and its
//                                                bias corrections -> me_adamw_ctl
//   adamw_seg_kernel                           : the AdamW pass with a per-segment (lr scale, weight decay) table in LDS
constexpr int GS_BLOCKS = 1024;
__global__ __launch_bounds__(EW_THREADS) void grad_stats_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
    // the squares are formed and summed in DOUBLE: the gradients are still multiplied by the loss scale here (the unscale is
    // part of adamw_prepare), and a finite gradient whose scaled square leaves the fp32 range must not read as non-finite --
    // GradScaler.unscale_ divides first and only then looks for inf (Video/utils.py:388-392).  HBM-bound either way.
    double ss = 0.0;
    float bad = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EW_THREADS) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool nf = (__float_as_uint(v[e]) & 0x7f800000u) == 0x7f800000u;            // inf or nan
            const double d = nf ? 0.0 : (double)v[e];                                        // (counted, not summed)
            ss = __builtin_fma(d, d, ss);
            bad += nf ? 1.f : 0.f;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
        const float v = g[n4 * 4 + threadIdx.x];
        const bool nf = (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u;
        const double d = nf ? 0.0 : (double)v;
        ss = __builtin_fma(d, d, ss);
        bad += nf ? 1.f : 0.f;
    }
    __shared__ double sh[2][EW_THREADS / 64];
    double dss = ss, dbad = bad;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dss += __shfl_xor(dss, o, 64); dbad += __shfl_xor(dbad, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = dss; sh[1][threadIdx.x >> 6] = dbad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
        for (int w = 0; w < EW_THREADS / 64; ++w) { a += sh[0][w]; b += sh[1][w]; }
        part[blockIdx.x * 2] = a;
        part[blockIdx.x * 2 + 1] = b;
    }
}
__global__ __launch_bounds__(64) void grad_stats_fold_kernel(const double* __restrict__ part, int nb, float* __restrict__ stats) {
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < nb; i += 64) { a += part[i * 2]; b += part[i * 2 + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    // stats[0] is the L2 NORM (not its square): representable in fp32 for any finite fp32 gradient buffer of < 2^64 elements
    if (threadIdx.x == 0) { stats[0] = (float)sqrt(a); stats[1] = (float)b; }
}
__global__ void adamw_prepare_kernel(me_adamw_ctl* __restrict__ ctl, const float* __restrict__ stats, const float* __restrict__ loss_scale,
                                     float grad_scale, float max_norm, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float mul = grad_scale;
    if (loss_scale) mul /= *loss_scale;                               // GradScaler.unscale_: grads were computed on loss * scale
    float norm = 0.f, bad = 0.f;
    if (stats) {
        norm = stats[0] * fabsf(mul);                                 // norm of the UNSCALED, averaged gradient
        bad = (stats[1] > 0.f || !(norm == norm) || norm > 3.0e38f) ? 1.f : 0.f;
        if (max_norm > 0.f) {
            // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped at 1
            const float coef = max_norm / (norm + 1e-6f);
            mul *= coef < 1.f ? coef : 1.f;
        }
    }
    int step = ctl->step;
    if (bad == 0.f) ++step;                                           // GradScaler.step skips optimizer.step(): no state moves
    ctl->step = step;
    ctl->grad_mul = mul;
    ctl->skip = bad;
    ctl->bc1 = 1.0f - powf(b1, (float)step);
    ctl->bc2_sqrt = sqrtf(1.0f - powf(b2, (float)step));
    ctl->total_norm = norm;
    ctl->found_inf = bad;
}
constexpr int ADAMW_MAX_SEG = 512;
__global__ __launch_bounds__(EW_THREADS) void adamw_seg_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                               float* __restrict__ v, int64_t n, const me_adamw_segment* __restrict__ seg,
                                                               int ns, float lr, float b1, float b2, float eps,
                                                               const me_adamw_ctl* __restrict__ ctl, bf16_t* __restrict__ mirror) {
    __shared__ int64_t s_end[ADAMW_MAX_SEG];
    __shared__ float s_lr[ADAMW_MAX_SEG], s_wd[ADAMW_MAX_SEG];
    for (int k = threadIdx.x; k < ns; k += EW_THREADS) { s_end[k] = seg[k].end; s_lr[k] = lr * seg[k].lr_scale; s_wd[k] = seg[k].weight_decay; }
    __syncthreads();
    const float gscale = ctl->grad_mul, bc1 = ctl->bc1, bc2_sqrt = ctl->bc2_sqrt;
    if (ctl->skip != 0.f) return;                                     // (uniform) non-finite gradient: parameters and moments stay
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
        int lo = 0, hi = ns - 1;                                      // the segment that holds i: first k with i < end[k]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (i < s_end[mid]) hi = mid; else lo = mid + 1;
        }
        const float lri = s_lr[lo], wd = s_wd[lo];
        float pi = p[i], mi = m[i], vi = v[i];
        adamw_update(pi, g[i] * gscale, mi, vi, lri, b1, b2, eps, wd, bc1, bc2_sqrt);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
        if (mirror) mirror[i] = (bf16_t)pi;
    }
}

// ---- position-embedding table resize (cls/pos-embed glue).
// Replaces TIMMVisionTransformer.resize_pos_embed (Image/detection/mmdet_custom/models/backbones/base/vit.py:459-486:
// reshape the [h*w, C] table to [1, C, h, w], F.interpolate(size=(H, W), mode, align_corners=False), flatten back) on the
// token-major layout directly: rows are grid positions, channels contiguous -- no permutes.  The sampling arithmetic is
// ATen's upsample_bicubic2d / upsample_bilinear2d restated: source coordinate s = (in/out) * (d + 0.5) - 0.5 (bilinear
// clamps it at 0, bicubic does not), cubic-convolution weights with A = -0.75, taps clamped to the table edge, fp32.
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__global__ __launch_bounds__(EW_THREADS) void resize_rows_kernel(const void* __restrict__ src, int src_dtype, void* __restrict__ dst,
                                                                 int dst_dtype, int h, int w, int H, int W, int C, int mode) {
    const int cq = C / 4;
    const int64_t total = (int64_t)H * W * cq;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int c = (int)(i % cq) * 4;
        const int ox = (int)((i / cq) % W), oy = (int)(i / ((int64_t)cq * W));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (mode == 1) {                 // bicubic
            const float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
            const float fly = floorf(fy), flx = floorf(fx);
            const int iy = (int)fly, ix = (int)flx;
            const float ty = fy - fly, tx = fx - flx;
            const float A = -0.75f;
            const float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
            const float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                int yy = iy - 1 + a;
                yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
                f32x4 row = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int xx = ix - 1 + b;
                    xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
                    row += wx[b] * load4_as_f32(src, src_dtype, ((int64_t)yy * w + xx) * C + c);
                }
                acc += wy[a] * row;
            }
        } else {                         // bilinear
            float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
            fy = fy < 0.f ? 0.f : fy;
            fx = fx < 0.f ? 0.f : fx;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 < h - 1 ? y0 + 1 : y0, x1 = x0 < w - 1 ? x0 + 1 : x0;
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const f32x4 v00 = load4_as_f32(src, src_dtype, ((int64_t)y0 * w + x0) * C + c);
            const f32x4 v01 = load4_as_f32(src, src_dtype, ((int64_t)y0 * w + x1) * C + c);
            const f32x4 v10 = load4_as_f32(src, src_dtype, ((int64_t)y1 * w + x0) * C + c);
            const f32x4 v11 = load4_as_f32(src, src_dtype, ((int64_t)y1 * w + x1) * C + c);
            acc = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        }
        store4_from_f32(dst, dst_dtype, ((int64_t)oy * W + ox) * C + c, acc);
    }
}

// ---- token pooling in front of the task heads: [B, N, C] tokens -> [B, C] features.
// Replaces x.mean(1) ahead of fc_norm (Video/models/modeling_finetune.py:445-454), x[:, 0] (cls token, same lines) and
// the 'max' / 'avg' global features of the PointCloud ClsHead (openpoints/models/classification/cls_base.py:126-133).
// One pass over the tokens, 4 channels per thread (coalesced along C), fp32 accumulation; max also records the arg-max
// token per channel (first index on ties, as torch.max) for the backward.
__global__ __launch_bounds__(EW_THREADS) void pool_tokens_kernel(const void* __restrict__ x, int x_dtype, float* __restrict__ out,
                                                                 int32_t* __restrict__ arg, int B, int N, int C, int mode) {
    const int cq = C / 4;
    const int64_t total = (int64_t)B * cq;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int c = (int)(i % cq) * 4, b = (int)(i / cq);
        const int64_t base = (int64_t)b * N * C + c;
        f32x4 acc = load4_as_f32(x, x_dtype, base);
        if (mode == ME_POOL_MEAN) {
            for (int n = 1; n < N; ++n) acc += load4_as_f32(x, x_dtype, base + (int64_t)n * C);
            acc *= 1.0f / (float)N;
        } else if (mode == ME_POOL_MAX) {
            int a[4] = {0, 0, 0, 0};
            for (int n = 1; n < N; ++n) {
                const f32x4 v = load4_as_f32(x, x_dtype, base + (int64_t)n * C);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > acc[e]) { acc[e] = v[e]; a[e] = n; }
            }
            if (arg) {
#pragma unroll
                for (int e = 0; e < 4; ++e) arg[(int64_t)b * C + c + e] = a[e];
            }
        }
        *reinterpret_cast<f32x4*>(out + (int64_t)b * C + c) = acc;
    }
}
__global__ __launch_bounds__(EW_THREADS) void pool_tokens_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ arg,
                                                                     void* __restrict__ dx, int dx_dtype, int B, int N, int C, int mode) {
    const int cq = C / 4;
    const int64_t total = (int64_t)B * N * cq;
    const float inv = 1.0f / (float)N;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int c = (int)(i % cq) * 4;
        const int n = (int)((i / cq) % N), b = (int)(i / ((int64_t)cq * N));
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (int64_t)b * C + c);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (mode == ME_POOL_MEAN) v = g * inv;
        else if (mode == ME_POOL_MAX) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = arg[(int64_t)b * C + c + e] == n ? g[e] : 0.f;
        } else if (n == 0) v = g;
        store4_from_f32(dx, dx_dtype, ((int64_t)b * N + n) * C + c, v);
    }
}

// ---- point-cloud tokenizer front end: farthest point sampling, k nearest neighbours, neighbourhood gather.
// Replaces, for PointPatchEmbed (PointCloud/openpoints/models/layers/group_embed.py:138-172):
//   furthest_point_sample  -> furthest_point_sampling_kernel (openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:101-210)
//   KNNGroup / KNN         -> torch.cdist + topk(largest=False) (openpoints/models/layers/group.py:12-28, 297-320)
//   grouping_operation + "relative_xyz"  (group.py:310-313)
// Index arithmetic is integer; the sampled / neighbour INDICES are what the parity tests compare bit-exactly.
//
// FPS.  What the reference kernel computes per round -- the next sample is the point that maximises min(d(p, last sample),
// temp[p]) -- is an arg-max with a SPECIFIC order among equal values, fixed by its thread-strided scan and its shared-memory
// tree (thread t = k mod T walks k = t, t + T, ... keeping the first strict maximum; partners (t, t + s), s = T/2 .. 1, fold
// with `v2 > v1 ? i2 : i1`): the winner is the candidate with the largest value, then the smallest BIT-REVERSED thread id,
// then the smallest k.  That is a total order, so here it is one 64-bit key per candidate
//     key = value bits (a non-negative float: monotone as an integer) << 32 | ~(bitrev(t) << ibits | k / T)
// and the arg-max is an unsigned maximum, taken in any association:
//   * one workgroup per cloud, T = the reference's thread count (opt_n_threads: largest power of two <= n, <= 1024), thread t
//     owns the same points k = t + i T as there -- but holds them, and their running minimum distances, in REGISTERS for all
//     m rounds (x, y, z, temp: 4 registers per point, up to 24 points per lane = 24 576 points; the reference re-reads
//     12 n bytes and read-modify-writes 4 n bytes of global memory every round);
//   * a lane's own scan is the reference's (first strict maximum in slot order); across the lanes of a wave the keys meet in
//     four DPP steps (quad_perm, row_half_mirror, row_mirror) + four v_readlane + scalar maxima, across the <= 16 waves in
//     one LDS word pair per wave and ONE barrier per round (slots alternate by round parity), against log2(T) + 1 barriers;
//   * larger clouds (n > 24 576) run the same rounds with the points and temp left in memory (FPS_MEM).
// The distance is evaluated exactly as hipcc evaluates the reference's expression (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) +
// (z2-z1)*(z2-z1) on gfx950 -- fma(dy, dy, dx*dx) + dz*dz, read off the ISA of the reference kernel built for this GPU -- so that the indices
// agree with the reference kernel built for this GPU bit for bit, not only on tie-rich lattices (tests/test_gpu_pointcloud_ref.py).
__device__ __forceinline__ float fps_dist(float dx, float dy, float dz) {
#pragma clang fp contract(off)
    const float xx = dx * dx, zz = dz * dz;
    const float xy = __builtin_fmaf(dy, dy, xx);
    return xy + zz;
}
template <int CTRL> __device__ __forceinline__ unsigned long long fps_dpp_max(unsigned long long v) {
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o > v ? o : v;
}
// maximum over the wave, wave-uniform result
__device__ __forceinline__ unsigned long long fps_wave_max(unsigned long long v) {
    v = fps_dpp_max<0xB1>(v);       // quad_perm [1,0,3,2]
    v = fps_dpp_max<0x4E>(v);       // quad_perm [2,3,0,1]
    v = fps_dpp_max<0x141>(v);      // row_half_mirror
    v = fps_dpp_max<0x140>(v);      // row_mirror: every lane of a row of 16 holds the row's maximum
    unsigned long long r = 0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned long long x = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), row * 16) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, row * 16);
        r = x > r ? x : r;
    }
    return r;
}
// R > 0: register-resident, R points per lane.  R == 0 (FPS_MEM): points and temp stay in memory.
template <int R>
__global__ __launch_bounds__(1024) void fps_kernel(const float* __restrict__ pts, float* __restrict__ temp, int32_t* __restrict__ idxs,
                                                   int n, int m, int logT, int ibits) {
    __shared__ unsigned long long wave_key[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = 1 << logT, nwaves = (int)(blockDim.x >> 6);
    pts += (int64_t)b * n * 3;
    temp += (int64_t)b * n;
    idxs += (int64_t)b * m;
    // ~(bitrev(t) << ibits): the thread part of the tie order (threads past T, in a wave that T does not fill, own nothing)
    const bool lane_on = tid < T;
    const unsigned rev = logT ? (__builtin_bitreverse32((unsigned)tid) >> (32 - logT)) : 0u;
    const unsigned tie_hi = ~(rev << ibits);
    constexpr int RR = R > 0 ? R : 1;
    float px[RR], py[RR], pz[RR], tmin[RR];
    if (R > 0) {
#pragma unroll
        for (int i = 0; i < RR; ++i) {
            // slots past the end of the cloud repeat point 0: their distance is 0 from the first round on, and a lane's scan keeps
            // the FIRST maximum, so they never displace a real point
            const int k = tid + i * T;
            const int kk = (lane_on && k < n) ? k : 0;
            px[i] = pts[kk * 3 + 0]; py[i] = pts[kk * 3 + 1]; pz[i] = pts[kk * 3 + 2];
            tmin[i] = 1e10f;          // the reference's caller fills temp with 1e10 (subsample.py: fill_(1e10))
        }
    } else {
        for (int k = tid; k < n; k += T) if (lane_on) temp[k] = 1e10f;
    }
    int old = 0;
    if (tid == 0) idxs[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];      // (wave-uniform address)
        float best = -1.f;
        unsigned bi = 0;
        if (R > 0) {
#pragma unroll
            for (int i = 0; i < RR; ++i) {
                const float d2 = fminf(fps_dist(px[i] - x1, py[i] - y1, pz[i] - z1), tmin[i]);
                tmin[i] = d2;
                bi = d2 > best ? (unsigned)i : bi;
                best = d2 > best ? d2 : best;
            }
        } else if (lane_on) {
            unsigned i = 0;
            for (int k = tid; k < n; k += T, ++i) {
                const float d2 = fminf(fps_dist(pts[k * 3 + 0] - x1, pts[k * 3 + 1] - y1, pts[k * 3 + 2] - z1), temp[k]);
                temp[k] = d2;
                bi = d2 > best ? i : bi;
                best = d2 > best ? d2 : best;
            }
        }
        unsigned long long key = lane_on ? (((unsigned long long)__float_as_uint(best) << 32) | (tie_hi - bi)) : 0ull;
        key = fps_wave_max(key);
        if (nwaves > 1) {
            if (lane == 0) wave_key[j & 1][wave] = key;
            __syncthreads();
            key = fps_wave_max(lane < nwaves ? wave_key[j & 1][lane] : 0ull);
        }
        const unsigned tie = ~(unsigned)key;                              // bitrev(t) << ibits | slot
        const unsigned t = logT ? (__builtin_bitreverse32(tie >> ibits) >> (32 - logT)) : 0u;
        old = (int)(((tie & ((1u << ibits) - 1u)) << logT) | t);
        old = __builtin_amdgcn_readfirstlane(old);
        if (tid == 0) idxs[j] = old;
    }
}

// k nearest support points of every query, ascending by squared distance, ties by lower index: one wave per query.
// Each lane scans its strided share keeping a sorted private top-k list is too register-hungry for k = 32; instead the
// wave selects the k winners one at a time (k rounds of a wave-wide arg-min over the distances held in LDS).
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ support, const float* __restrict__ query,
                                                  int32_t* __restrict__ idx, int n, int m, int k) {
    extern __shared__ __attribute__((aligned(16))) char knn_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* dist = reinterpret_cast<float*>(knn_smem) + (int64_t)wave * n;      // this wave's [n] distances
    const int b = blockIdx.y;
    const int q = blockIdx.x * 4 + wave;
    if (q >= m) return;                                                         // (whole wave)
    support += (int64_t)b * n * 3;
    const float qx = query[((int64_t)b * m + q) * 3 + 0], qy = query[((int64_t)b * m + q) * 3 + 1], qz = query[((int64_t)b * m + q) * 3 + 2];
    for (int i = lane; i < n; i += 64) {
        const float dx = support[i * 3 + 0] - qx, dy = support[i * 3 + 1] - qy, dz = support[i * 3 + 2] - qz;
        dist[i] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    }
    __builtin_amdgcn_wave_barrier();
    int32_t* out = idx + ((int64_t)b * m + q) * k;
    for (int r = 0; r < k; ++r) {
        float best = 3.0e38f;
        int besti = 0x7fffffff;
        for (int i = lane; i < n; i += 64) {
            const float d = dist[i];
            if (d < best) { best = d; besti = i; }                              // first (lowest index) of equal values per lane
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(besti, o, 64);
            if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (lane == 0) { out[r] = besti; dist[besti] = 3.0e38f; }
        __builtin_amdgcn_wave_barrier();
    }
}

// rows[(b, s, j), 0..2] = pts[b, idx[b, s, j]] - centers[b, s] (relative_xyz), columns 3 .. cols-1 zero: the first MLP layer's
// GEMM operand (reduction dimension padded to the GEMM's 16-byte granule).
__global__ __launch_bounds__(EW_THREADS) void group_rel_kernel(const float* __restrict__ pts, const float* __restrict__ centers,
                                                               const int32_t* __restrict__ idx, float* __restrict__ rows, int B, int n,
                                                               int m, int k, int cols) {
    const int64_t total = (int64_t)B * m * k;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t bs = i / k;
        const int b = (int)(bs / m);
        const int id = idx[i];
        const float* p = pts + ((int64_t)b * n + id) * 3;
        const float* c = centers + bs * 3;
        float* r = rows + i * cols;
        r[0] = p[0] - c[0]; r[1] = p[1] - c[1]; r[2] = p[2] - c[2];
        for (int j = 3; j < cols; ++j) r[j] = 0.f;
    }
}

}  // namespace

extern "C" int me_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(src && dst && n >= 0, "me_cast: bad args");
    ME_CHECK_ARG(me_storage_dtype_ok(src_dtype) && me_storage_dtype_ok(dst_dtype), "me_cast: bad dtype");
    if (n == 0) return ME_OK;
    if (src_dtype == ME_F16 || dst_dtype == ME_F16) {
        hipLaunchKernelGGL(cast_any_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream, src, src_dtype, dst, dst_dtype, n);
        ME_CHECK_LAUNCH("me_cast");
        return ME_OK;
    }
    hipLaunchKernelGGL(cast_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(EW_THREADS), 0, stream, src, src_dtype, dst, dst_dtype, n);
    ME_CHECK_LAUNCH("me_cast");
    return ME_OK;
}

extern "C" int me_split3(const float* src, int64_t ld_src, void* dst, int64_t rows, int64_t cols, int right_operand, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(src && dst && rows >= 0 && cols > 0 && cols % 4 == 0 && ld_src >= cols && ld_src % 4 == 0, "me_split3: bad args");
    ME_CHECK_ARG(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0), "me_split3: src must be 16-byte, dst 8-byte aligned");
    if (rows == 0) return ME_OK;
    hipLaunchKernelGGL(split3_kernel, dim3(ew_blocks(rows * (cols / 4))), dim3(EW_THREADS), 0, stream, src, ld_src,
                       reinterpret_cast<uint16_t*>(dst), rows, cols, right_operand);
    ME_CHECK_LAUNCH("me_split3");
    return ME_OK;
}

extern "C" int me_transpose_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t rows, int64_t cols,
                                 void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(src && dst && rows > 0 && cols > 0, "me_transpose_cast: bad args");
    ME_CHECK_ARG(me_storage_dtype_ok(src_dtype) && me_storage_dtype_ok(dst_dtype), "me_transpose_cast: bad dtype");
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL(transpose_cast_kernel, grid, dim3(256), 0, stream, src, src_dtype, dst, dst_dtype, rows, cols);
    ME_CHECK_LAUNCH("me_transpose_cast");
    return ME_OK;
}

extern "C" int me_transpose_cast_batched(const me_tc_batch* b, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(b && b->n >= 0 && b->n <= ME_TC_BATCH, "me_transpose_cast_batched: bad batch");
    ME_CHECK_ARG(me_storage_dtype_ok(b->src_dtype) && me_storage_dtype_ok(b->dst_dtype), "me_transpose_cast_batched: bad dtype");
    int64_t tiles = 0;
    for (int k = 0; k < b->n; ++k) {
        ME_CHECK_ARG(b->item[k].src && b->item[k].dst && b->item[k].rows > 0 && b->item[k].cols > 0,
                     "me_transpose_cast_batched: bad item %d", k);
        tiles += ((b->item[k].cols + 63) / 64) * ((b->item[k].rows + 63) / 64);
    }
    if (tiles == 0) return ME_OK;
    ME_CHECK_ARG(tiles < (1ll << 31), "me_transpose_cast_batched: too many tiles");
    hipLaunchKernelGGL(transpose_cast_batched_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, *b);
    ME_CHECK_LAUNCH("me_transpose_cast_batched");
    return ME_OK;
}

extern "C" int me_split3_batched(const me_tc_batch* b, int transposed, int right_operand, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(b && b->n >= 0 && b->n <= ME_TC_BATCH, "me_split3_batched: bad batch");
    ME_CHECK_ARG(b->src_dtype == ME_F32 && b->dst_dtype == ME_BF16X3, "me_split3_batched: fp32 sources, ME_BF16X3 destinations");
    int64_t tiles = 0;
    for (int k = 0; k < b->n; ++k) {
        ME_CHECK_ARG(b->item[k].src && b->item[k].dst && b->item[k].rows > 0 && b->item[k].cols > 0 && b->item[k].rows % 4 == 0 &&
                         b->item[k].cols % 4 == 0 && (uintptr_t)b->item[k].src % 16 == 0 && (uintptr_t)b->item[k].dst % 8 == 0,
                     "me_split3_batched: bad item %d (rows, cols multiples of 4; src 16-byte, dst 8-byte aligned)", k);
        tiles += ((b->item[k].cols + 63) / 64) * ((b->item[k].rows + 63) / 64);
    }
    if (tiles == 0) return ME_OK;
    ME_CHECK_ARG(tiles < (1ll << 31), "me_split3_batched: too many tiles");
    hipLaunchKernelGGL(split3_batched_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, *b, transposed, right_operand);
    ME_CHECK_LAUNCH("me_split3_batched");
    return ME_OK;
}

extern "C" int me_add_rows(const void* x, int x_dtype, const void* pos, int pos_dtype, void* y, int y_dtype, int64_t rows,
                           int64_t pos_rows, int cols, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(x && pos && y && rows > 0 && pos_rows > 0 && cols > 0 && cols % 4 == 0, "me_add_rows: bad args");
    ME_CHECK_ARG(me_dtype_ok(x_dtype) && me_dtype_ok(pos_dtype) && me_dtype_ok(y_dtype), "me_add_rows: bad dtype");
    hipLaunchKernelGGL(add_rows_kernel, dim3(ew_blocks(rows * (cols / 4))), dim3(EW_THREADS), 0, stream, x, x_dtype, pos,
                       pos_dtype, y, y_dtype, rows, pos_rows, cols);
    ME_CHECK_LAUNCH("me_add_rows");
    return ME_OK;
}

extern "C" int me_resize_rows(const void* src, int src_dtype, void* dst, int dst_dtype, int h, int w, int H, int W, int cols,
                              int mode, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(src && dst && h > 0 && w > 0 && H > 0 && W > 0 && cols > 0 && cols % 4 == 0, "me_resize_rows: bad args");
    ME_CHECK_ARG(me_dtype_ok(src_dtype) && me_dtype_ok(dst_dtype), "me_resize_rows: bad dtype");
    ME_CHECK_ARG(mode == ME_RESIZE_BILINEAR || mode == ME_RESIZE_BICUBIC, "me_resize_rows: mode %d (bilinear / bicubic only)", mode);
    hipLaunchKernelGGL(resize_rows_kernel, dim3(ew_blocks((int64_t)H * W * (cols / 4))), dim3(EW_THREADS), 0, stream, src,
                       src_dtype, dst, dst_dtype, h, w, H, W, cols, mode);
    ME_CHECK_LAUNCH("me_resize_rows");
    return ME_OK;
}

extern "C" int me_pool_tokens(const void* x, int x_dtype, float* out, int32_t* argmax, int B, int N, int C, int mode, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(x && out && B > 0 && N > 0 && C > 0 && C % 4 == 0, "me_pool_tokens: bad args");
    ME_CHECK_ARG(me_dtype_ok(x_dtype), "me_pool_tokens: bad dtype");
    ME_CHECK_ARG(mode == ME_POOL_MEAN || mode == ME_POOL_MAX || mode == ME_POOL_FIRST, "me_pool_tokens: bad mode %d", mode);
    hipLaunchKernelGGL(pool_tokens_kernel, dim3(ew_blocks((int64_t)B * (C / 4))), dim3(EW_THREADS), 0, stream, x, x_dtype, out,
                       argmax, B, N, C, mode);
    ME_CHECK_LAUNCH("me_pool_tokens");
    return ME_OK;
}

extern "C" int me_pool_tokens_bwd(const float* dy, const int32_t* argmax, void* dx, int dx_dtype, int B, int N, int C, int mode,
                                  void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(dy && dx && B > 0 && N > 0 && C > 0 && C % 4 == 0, "me_pool_tokens_bwd: bad args");
    ME_CHECK_ARG(me_dtype_ok(dx_dtype), "me_pool_tokens_bwd: bad dtype");
    ME_CHECK_ARG(mode == ME_POOL_MEAN || mode == ME_POOL_FIRST || (mode == ME_POOL_MAX && argmax),
                 "me_pool_tokens_bwd: bad mode %d (max needs the forward's argmax)", mode);
    hipLaunchKernelGGL(pool_tokens_bwd_kernel, dim3(ew_blocks((int64_t)B * N * (C / 4))), dim3(EW_THREADS), 0, stream, dy, argmax, dx,
                       dx_dtype, B, N, C, mode);
    ME_CHECK_LAUNCH("me_pool_tokens_bwd");
    return ME_OK;
}

extern "C" int me_fps(const float* points, int32_t* idx, float* temp, int B, int n, int m, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(points && idx && temp && B > 0 && n > 0 && m > 0 && m <= n, "me_fps: bad args (B=%d n=%d m=%d)", B, n, m);
    int logT = 0;
    while ((2 << logT) <= n && logT < 10) ++logT;          // opt_n_threads(n) of the reference launcher: T = 1 << logT
    const int T = 1 << logT, threads = T < 64 ? 64 : T;
    const int per_lane = (n + T - 1) / T;                   // points per reference thread
    int ibits = 1;
    while ((1 << ibits) < per_lane) ++ibits;
#define ME_FPS_CASE(RR) hipLaunchKernelGGL(fps_kernel<RR>, dim3((unsigned)B), dim3((unsigned)threads), 0, stream, points, temp, idx, n, m, logT, ibits)
    if (per_lane <= 1) ME_FPS_CASE(1);
    else if (per_lane <= 2) ME_FPS_CASE(2);
    else if (per_lane <= 4) ME_FPS_CASE(4);
    else if (per_lane <= 8) ME_FPS_CASE(8);
    else if (per_lane <= 16) ME_FPS_CASE(16);
    else if (per_lane <= 24) ME_FPS_CASE(24);
    else ME_FPS_CASE(0);
#undef ME_FPS_CASE
    ME_CHECK_LAUNCH("me_fps");
    return ME_OK;
}

extern "C" int me_knn(const float* support, const float* query, int32_t* idx, int B, int n, int m, int k, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(support && query && idx && B > 0 && n > 0 && m > 0 && k > 0 && k <= n, "me_knn: bad args");
    const size_t lds = (size_t)4 * n * sizeof(float);
    ME_CHECK_ARG(lds <= 160 * 1024, "me_knn: %d support points exceed the LDS-resident form (max 10240)", n);
    static OncePerDevice once;
    if (once.need()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(knn_kernel, dim3((unsigned)((m + 3) / 4), (unsigned)B), dim3(256), lds, stream, support, query, idx, n, m, k);
    ME_CHECK_LAUNCH("me_knn");
    return ME_OK;
}

extern "C" int me_group_relative(const float* points, const float* centers, const int32_t* idx, float* rows, int B, int n, int m,
                                 int k, int cols, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(points && centers && idx && rows && B > 0 && n > 0 && m > 0 && k > 0 && cols >= 3, "me_group_relative: bad args");
    hipLaunchKernelGGL(group_rel_kernel, dim3(ew_blocks((int64_t)B * m * k)), dim3(EW_THREADS), 0, stream, points, centers, idx,
                       rows, B, n, m, k, cols);
    ME_CHECK_LAUNCH("me_group_relative");
    return ME_OK;
}

extern "C" int me_window_rows(const void* src, void* dst, int dtype, int B, int H, int W, int window, int cols, int merge,
                              void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && window > 0 && cols > 0, "me_window_rows: bad args");
    ME_CHECK_ARG(me_dtype_ok(dtype), "me_window_rows: bad dtype");
    const int64_t row_bytes = (int64_t)cols * (int64_t)me_dtype_size(dtype);
    ME_CHECK_ARG(row_bytes % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0,
                 "me_window_rows: rows must be multiples of 16 bytes, 16-byte aligned");
    const int chunks = (int)(row_bytes / 16);
    const int gh = (H + window - 1) / window, gw = (W + window - 1) / window;
    const int64_t rows = merge ? (int64_t)B * H * W : (int64_t)B * gh * gw * window * window;
    hipLaunchKernelGGL(window_rows_kernel, dim3(ew_blocks(rows * chunks)), dim3(EW_THREADS), 0, stream,
                       reinterpret_cast<const u32x4*>(src), reinterpret_cast<u32x4*>(dst), B, H, W, window, chunks, merge);
    ME_CHECK_LAUNCH("me_window_rows");
    return ME_OK;
}

extern "C" size_t me_colsum_workspace(int64_t cols) { return (size_t)CS_SPLITS * (size_t)cols * sizeof(float); }

extern "C" int me_colsum(const void* x, int x_dtype, int64_t ldx, int64_t rows, int64_t cols, float* out, int accumulate,
                         void* workspace, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(x && out && workspace && rows > 0 && cols > 0, "me_colsum: bad args");
    ME_CHECK_ARG(me_dtype_ok(x_dtype), "me_colsum: bad dtype");
    float* partial = reinterpret_cast<float*>(workspace);
    if (cols % 4 == 0 && ldx % 4 == 0 && (uintptr_t)x % 16 == 0) {
        dim3 grid((unsigned)((cols + 255) / 256), CS_SPLITS);
        if (x_dtype == ME_BF16)
            hipLaunchKernelGGL(colsum_partial_vec_kernel<bf16_t>, grid, dim3(256), 0, stream, x, ldx, rows, cols, partial);
        else
            hipLaunchKernelGGL(colsum_partial_vec_kernel<float>, grid, dim3(256), 0, stream, x, ldx, rows, cols, partial);
    } else {
        dim3 grid((unsigned)((cols + 63) / 64), CS_SPLITS);
        hipLaunchKernelGGL(colsum_partial_kernel, grid, dim3(256), 0, stream, x, x_dtype, ldx, rows, cols, partial);
    }
    ME_CHECK_LAUNCH("me_colsum(partial)");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, partial, cols, out,
                       accumulate);
    ME_CHECK_LAUNCH("me_colsum(final)");
    return ME_OK;
}

extern "C" int me_colsum_mul(const void* x, int x_dtype, int64_t ldx, const void* y, int y_dtype, int64_t ldy, int64_t rows,
                             int64_t cols, float* out, int accumulate, void* workspace, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(x && y && out && workspace && rows > 0 && cols > 0, "me_colsum_mul: bad args");
    ME_CHECK_ARG(me_dtype_ok(x_dtype) && me_dtype_ok(y_dtype), "me_colsum_mul: bad dtype");
    ME_CHECK_ARG(cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "me_colsum_mul: cols and row strides must be multiples of 4");
    float* partial = reinterpret_cast<float*>(workspace);
    dim3 grid((unsigned)((cols + 255) / 256), CS_SPLITS);
    hipLaunchKernelGGL(colsum_mul_partial_kernel, grid, dim3(256), 0, stream, x, x_dtype, ldx, y, y_dtype, ldy, rows, cols, partial);
    ME_CHECK_LAUNCH("me_colsum_mul(partial)");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, partial, cols, out,
                       accumulate);
    ME_CHECK_LAUNCH("me_colsum_mul(final)");
    return ME_OK;
}

// non-overlapping patches (stride >= kernel on every axis), kw % 4 == 0, aligned quads: the scatter four pixels at a time
__global__ __launch_bounds__(EW_THREADS) void unpatchify_add4_kernel(const void* __restrict__ dcols, int ddt, float* __restrict__ dx_out, PatchGeom g) {
    const int64_t feat4 = (int64_t)g.Cin * g.kt * g.kh * (g.kw / 4);
    const int64_t ntok = (int64_t)g.B * g.gt * g.gh * g.gw;
    const int64_t total = ntok * feat4;
    for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
        const int64_t tok = i / feat4;
        int64_t f = i - tok * feat4;
        const int dx = (int)(f % (g.kw / 4)) * 4; f /= (g.kw / 4);
        const int dy = (int)(f % g.kh); f /= g.kh;
        const int dt = (int)(f % g.kt); f /= g.kt;
        const int c = (int)f;
        int64_t t = tok;
        const int px = (int)(t % g.gw); t /= g.gw;
        const int py = (int)(t % g.gh); t /= g.gh;
        const int pt = (int)(t % g.gt); t /= g.gt;
        const int b = (int)t;
        const int64_t dst = ((((int64_t)b * g.Cin + c) * g.T + (pt * g.st + dt)) * g.H + (py * g.sh + dy)) * g.W + (px * g.sw + dx);
        f32x4* q = reinterpret_cast<f32x4*>(dx_out + dst);
        *q = *q + load4_as_f32(dcols, ddt, i * 4);
    }
}

static int make_geom(PatchGeom& g, int B, int Cin, int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw) {
    ME_CHECK_ARG(B > 0 && Cin > 0 && T > 0 && H > 0 && W > 0 && kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0,
                 "patchify: bad geometry");
    ME_CHECK_ARG(T >= kt && H >= kh && W >= kw, "patchify: kernel larger than input");
    g = PatchGeom{B, Cin, T, H, W, kt, kh, kw, st, sh, sw, (T - kt) / st + 1, (H - kh) / sh + 1, (W - kw) / sw + 1};
    return ME_OK;
}

extern "C" int me_patchify(const void* x, int x_dtype, void* cols, int cols_dtype, int B, int Cin, int T, int H, int W,
                           int kt, int kh, int kw, int st, int sh, int sw, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(x && cols, "me_patchify: null pointer");
    ME_CHECK_ARG(me_dtype_ok(x_dtype) && me_dtype_ok(cols_dtype), "me_patchify: bad dtype");
    PatchGeom g;
    int rc = make_geom(g, B, Cin, T, H, W, kt, kh, kw, st, sh, sw);
    if (rc) return rc;
    const int64_t total = (int64_t)B * g.gt * g.gh * g.gw * Cin * kt * kh * kw;
    if (kw % 4 == 0 && (uintptr_t)cols % 16 == 0) {
        // (source quads aligned: every row start and every patch column offset a multiple of 4 elements, the base 16-byte aligned)
        const int aligned = (W % 4 == 0 && sw % 4 == 0 && (uintptr_t)x % 16 == 0) ? 1 : 0;
        hipLaunchKernelGGL(patchify4_kernel, dim3(ew_blocks(total / 4)), dim3(EW_THREADS), 0, stream, x, x_dtype, cols, cols_dtype, g, aligned);
    } else {
        hipLaunchKernelGGL(patchify_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, stream, x, x_dtype, cols, cols_dtype, g);
    }
    ME_CHECK_LAUNCH("me_patchify");
    return ME_OK;
}

extern "C" int me_unpatchify_add(const void* dcols, int dcols_dtype, float* dx, int B, int Cin, int T, int H, int W, int kt,
                                 int kh, int kw, int st, int sh, int sw, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(dcols && dx, "me_unpatchify_add: null pointer");
    ME_CHECK_ARG(me_dtype_ok(dcols_dtype), "me_unpatchify_add: bad dtype");
    PatchGeom g;
    int rc = make_geom(g, B, Cin, T, H, W, kt, kh, kw, st, sh, sw);
    if (rc) return rc;
    const int64_t total = (int64_t)B * g.gt * g.gh * g.gw * Cin * kt * kh * kw;
    const bool overlap = st < kt || sh < kh || sw < kw;
    if (!overlap && kw % 4 == 0 && W % 4 == 0 && sw % 4 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)dcols % 16 == 0) {
        hipLaunchKernelGGL(unpatchify_add4_kernel, dim3(ew_blocks(total / 4)), dim3(EW_THREADS), 0, stream, dcols, dcols_dtype, dx, g);
        ME_CHECK_LAUNCH("me_unpatchify_add");
        return ME_OK;
    }
    hipLaunchKernelGGL(unpatchify_add_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, stream, dcols, dcols_dtype, dx, g);
    ME_CHECK_LAUNCH("me_unpatchify_add");
    return ME_OK;
}

// circular unfold for the Conv1d weight gradient: out[(b,l), i*3 + k] = x[b, (l + k - 1) mod L, i], zero in the pad columns
__global__ __launch_bounds__(EW_THREADS) void ts_unfold_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int L,
                                                               int cin, int ncols) {
    const int64_t total = (int64_t)B * L * ncols;
    for (int64_t t = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; t < total; t += (int64_t)gridDim.x * EW_THREADS) {
        const int col = (int)(t % ncols);
        const int64_t row = t / ncols;
        float v = 0.f;
        if (col < 3 * cin) {
            const int i = col / 3, k = col % 3;
            const int l = (int)(row % L);
            const int64_t b = row / L;
            int ls = l + k - 1;
            ls = ls < 0 ? ls + L : (ls >= L ? ls - L : ls);
            v = x[(b * L + ls) * cin + i];
        }
        out[t] = v;
    }
}

extern "C" int me_timeseries_unfold(const float* x, float* out, int B, int L, int cin, int ncols, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(x && out && B > 0 && L > 0 && cin > 0 && ncols >= 3 * cin, "me_timeseries_unfold: bad args");
    hipLaunchKernelGGL(ts_unfold_kernel, dim3(ew_blocks((int64_t)B * L * ncols)), dim3(EW_THREADS), 0, stream, x, out, B, L, cin,
                       ncols);
    ME_CHECK_LAUNCH("me_timeseries_unfold");
    return ME_OK;
}

extern "C" int me_timeseries_embed(const float* x, const float* conv_w, const int32_t* marks, int n_mark,
                                   const float* const* tables, const int32_t* table_rows, const float* pe, void* out,
                                   int out_dtype, int B, int L, int cin, int C, int32_t* err_flag, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(x && conv_w && out && B > 0 && L > 0 && cin > 0 && C > 0, "me_timeseries_embed: bad args");
    ME_CHECK_ARG(me_dtype_ok(out_dtype), "me_timeseries_embed: bad dtype");
    ME_CHECK_ARG(n_mark >= 0 && n_mark <= 5, "me_timeseries_embed: n_mark must be 0..5");
    ME_CHECK_ARG(n_mark == 0 || (marks && tables && table_rows), "me_timeseries_embed: marks given without tables");
    TsTables tt;
    for (int f = 0; f < TS_MAX_MARK; ++f) { tt.tab[f] = nullptr; tt.rows[f] = 0; }
    for (int f = 0; f < n_mark; ++f) {   // `tables` / `table_rows` are HOST arrays of device pointers / sizes
        tt.tab[f] = tables[f];
        tt.rows[f] = table_rows[f];
        ME_CHECK_ARG(tt.tab[f] && tt.rows[f] > 0, "me_timeseries_embed: bad table %d", f);
    }
    hipLaunchKernelGGL(ts_embed_kernel, dim3((unsigned)((int64_t)B * L)), dim3(EW_THREADS), 0, stream, x, conv_w,
                       n_mark ? marks : nullptr, n_mark, tt, pe, out, out_dtype, B, L, cin, C, err_flag);
    ME_CHECK_LAUNCH("me_timeseries_embed");
    return ME_OK;
}

extern "C" int me_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             void* bf16_mirror, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "me_adamw_step: bad args");
    if (n == 0) return ME_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    const bool vec = n >= 4 && (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                     ((uintptr_t)bf16_mirror & 7) == 0;
    if (vec)
        hipLaunchKernelGGL(adamw_kernel<true>, dim3(ew_blocks(n / 4)), dim3(EW_THREADS), 0, stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                           beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale, reinterpret_cast<bf16_t*>(bf16_mirror));
    else
        hipLaunchKernelGGL(adamw_kernel<false>, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                           beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale, reinterpret_cast<bf16_t*>(bf16_mirror));
    ME_CHECK_LAUNCH("me_adamw_step");
    return ME_OK;
}

extern "C" size_t me_grad_stats_workspace(void) { return (size_t)GS_BLOCKS * 2 * sizeof(double); }

extern "C" int me_grad_stats(const float* grad, int64_t n, float* stats, void* workspace, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(grad && stats && workspace && n > 0, "me_grad_stats: bad args");
    ME_CHECK_ARG((uintptr_t)grad % 16 == 0, "me_grad_stats: the gradient buffer must be 16-byte aligned");
    int nb = ew_blocks(n / 4 + 1);
    nb = nb < GS_BLOCKS ? nb : GS_BLOCKS;
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(grad_stats_kernel, dim3((unsigned)nb), dim3(EW_THREADS), 0, stream, grad, n, part);
    hipLaunchKernelGGL(grad_stats_fold_kernel, dim3(1), dim3(64), 0, stream, part, nb, stats);
    ME_CHECK_LAUNCH("me_grad_stats");
    return ME_OK;
}

extern "C" int me_adamw_prepare(me_adamw_ctl* ctl, const float* stats, const float* loss_scale, float grad_scale, float max_norm,
                                float beta1, float beta2, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(ctl != nullptr, "me_adamw_prepare: null control block");
    hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(1), 0, stream, ctl, stats, loss_scale, grad_scale, max_norm, beta1, beta2);
    ME_CHECK_LAUNCH("me_adamw_prepare");
    return ME_OK;
}

extern "C" int me_adamw_step_segments(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                      const me_adamw_segment* segments, int n_segments, float lr, float beta1, float beta2, float eps,
                                      const me_adamw_ctl* ctl, void* bf16_mirror, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && segments && ctl && n >= 0, "me_adamw_step_segments: bad args");
    ME_CHECK_ARG(n_segments >= 1 && n_segments <= ADAMW_MAX_SEG, "me_adamw_step_segments: 1 .. %d segments (got %d)", ADAMW_MAX_SEG, n_segments);
    if (n == 0) return ME_OK;
    hipLaunchKernelGGL(adamw_seg_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, stream, param, grad, exp_avg, exp_avg_sq, n, segments,
                       n_segments, lr, beta1, beta2, eps, ctl, reinterpret_cast<bf16_t*>(bf16_mirror));
    ME_CHECK_LAUNCH("me_adamw_step_segments");
    return ME_OK;
}

extern "C" int me_dropout_add(const void* v, int v_dtype, const void* res, int res_dtype, void* out, int out_dtype,
                              int64_t rows, int cols, int64_t rows_per_sample, float p_drop, float p_path, uint64_t seed, const float* colscale,
                              void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ME_CHECK_ARG(v && out && rows > 0 && cols > 0 && cols % 4 == 0 && rows_per_sample > 0, "me_dropout_add: bad args");
    ME_CHECK_ARG(me_dtype_ok(v_dtype) && me_dtype_ok(out_dtype) && (!res || me_dtype_ok(res_dtype)), "me_dropout_add: bad dtype");
    ME_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && p_path >= 0.f && p_path < 1.f, "me_dropout_add: probabilities must be in [0, 1)");
    hipLaunchKernelGGL(dropout_add_kernel, dim3(ew_blocks(rows * (cols / 4))), dim3(EW_THREADS), 0, stream, v, v_dtype, res,
                       res_dtype, out, out_dtype, rows, cols, rows_per_sample, p_drop, p_path, seed, colscale);
    ME_CHECK_LAUNCH("me_dropout_add");
    return ME_OK;
}
